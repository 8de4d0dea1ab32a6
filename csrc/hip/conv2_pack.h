// bf16 copies of the MNIST conv2 filter W[800 = tap*32+cin][64 cout] (fp32 master weights), laid out exactly as the two
// LDS-resident conv2 kernels keep them, so their filter staging is a straight 16-byte copy of half the bytes with no
// conversion (it was a third to a half of those kernels' time: 100-200 KB of fp32 per workgroup through L2):
//   Wf[half][800][32]  : the forward kernel's slice for output channels [32*half, 32*half+32), 64-byte rows
//   Wb[800][64]        : the data-gradient kernel's image, 128-byte rows whose 16-byte chunks are XOR-swizzled with row&7
// The step executor packs them with a few extra workgroups of the conv1 forward launch (first kernel of a step, does not
// touch conv2's weights), i.e. after the previous step's optimizer update and at no extra launch.
#pragma once
#include "common.h"

constexpr int CONV2_WF_BYTES = 2 * 800 * 64;
constexpr int CONV2_WB_BYTES = 800 * 128;

__device__ __forceinline__ void conv2_pack_body(const float* __restrict__ W, char* __restrict__ Wf, char* __restrict__ Wb,
                                                const int block, const int nblocks) {
  for (int c = block * 256 + threadIdx.x; c < 6400; c += nblocks * 256) {  // 16-byte output chunk = 8 output channels
    const f32x4 v0 = reinterpret_cast<const f32x4*>(W)[2 * c], v1 = reinterpret_cast<const f32x4*>(W)[2 * c + 1];
    const int row = c >> 3, ch = c & 7;
    const uint4 pk = make_uint4(pack_bf2(v0[0], v0[1]), pack_bf2(v0[2], v0[3]), pack_bf2(v1[0], v1[1]), pack_bf2(v1[2], v1[3]));
    *reinterpret_cast<uint4*>(Wf + (ch >> 2) * (800 * 64) + row * 64 + (ch & 3) * 16) = pk;
    *reinterpret_cast<uint4*>(Wb + row * 128 + ((ch ^ (row & 7)) << 4)) = pk;
  }
}
