// Three-plane bf16 copies (bf16x3.h) of the MNIST conv2 filter W[25 taps][32 cin][64 cout] (fp32 master weights) for the
// "bf16x6" conv2 kernels (f32x3_conv2.hip), rewritten by leading workgroups of the conv1 forward launch like the one-plane
// copies of conv2_pack.h:
//   Wf3[plane][tap][cout 64][cin 32]  : the forward's B operand (k = cin runs fastest: one 16-byte load = a lane's 8 k values)
//   Wb3[plane][tap][cin 32][cout 64]  : the data gradient's B operand (k = cout runs fastest; W's own layout)
#pragma once
#include "bf16x3.h"

constexpr int CONV2_W3_PLANE = 25 * 32 * 64 * 2;  // bytes per plane
constexpr int CONV2_W3_BYTES = 3 * CONV2_W3_PLANE;

__device__ __forceinline__ void conv2_pack3_body(const float* __restrict__ W, char* __restrict__ Wf3, char* __restrict__ Wb3,
                                                 const int block, const int nblocks) {
  // forward copy: chunk = (tap, cin group of 8, cout), cout fastest so the strided reads coalesce
  for (int c = block * 256 + threadIdx.x; c < 6400; c += nblocks * 256) {
    const int n = c & 63, kg = (c >> 6) & 3, t = c >> 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(long)(t * 32 + 8 * kg + j) * 64 + n];
    const Bf3Quad lo = bf3_split4(f32x4{v[0], v[1], v[2], v[3]}), hi = bf3_split4(f32x4{v[4], v[5], v[6], v[7]});
    char* dst = Wf3 + ((long)(t * 64 + n) * 32 + 8 * kg) * 2;
    *reinterpret_cast<uint4*>(dst) = make_uint4(lo.p0.x, lo.p0.y, hi.p0.x, hi.p0.y);
    *reinterpret_cast<uint4*>(dst + CONV2_W3_PLANE) = make_uint4(lo.p1.x, lo.p1.y, hi.p1.x, hi.p1.y);
    *reinterpret_cast<uint4*>(dst + 2 * CONV2_W3_PLANE) = make_uint4(lo.p2.x, lo.p2.y, hi.p2.x, hi.p2.y);
  }
  // data-gradient copy: W's layout, 4 values per chunk
  for (int c = block * 256 + threadIdx.x; c < 12800; c += nblocks * 256) {
    const Bf3Quad s = bf3_split4(reinterpret_cast<const f32x4*>(W)[c]);
    char* dst = Wb3 + (long)c * 8;
    *reinterpret_cast<uint2*>(dst) = s.p0;
    *reinterpret_cast<uint2*>(dst + CONV2_W3_PLANE) = s.p1;
    *reinterpret_cast<uint2*>(dst + 2 * CONV2_W3_PLANE) = s.p2;
  }
}
