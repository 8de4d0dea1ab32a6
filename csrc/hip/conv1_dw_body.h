// conv1 filter/bias gradient of the MNIST CNN as a device function, so it can be launched on its own
// (mnist_kernels.hip) or horizontally fused with the conv2 filter gradient (conv2_lds.hip): the two are
// independent, each fills well under half of the chip, and a kernel boundary costs more than either saves.
#pragma once
#include "common.h"

// index of the sample feeding batch row b of the current step
__device__ __forceinline__ int sample_index(const int* perm, const int* step_ptr, int spe, int B, int b) {
  if (perm == nullptr) return b;
  const int step = step_ptr ? *step_ptr : 0;
  return perm[(long)(step % spe) * B + b];
}

// LDS carve-up (floats): x halo plane 32x32 | gradient plane 7*256*4 (196*32 used) | argmax plane 4*256*8 bytes |
// cross-group reduction 8*26*32
constexpr int CONV1_DW_LDS_FLOATS = 1024 + 7168 + 2048 + 6656;

// dpool1: fp32 [B,14,14,32] gradient wrt the pooled+ReLU'd conv1 output (already ReLU-masked); amax: pool argmax.
// dW[25][32], db[32] accumulated with atomics (pre-zeroed).  Workgroup `block` of `nblocks` strides over the images.
// Threads 0..255 work; a larger workgroup's extra threads only take part in the barriers.
__device__ __forceinline__ void conv1_dw_body(float* __restrict__ smem, const int block, const int nblocks,
                                              const float* __restrict__ images, const int* __restrict__ perm,
                                              const int* __restrict__ step_ptr, int spe, int B,
                                              const float* __restrict__ dpool1, const uint8_t* __restrict__ amax,
                                              float* __restrict__ dW, float* __restrict__ db) {
  // per image: x halo plane (4 KB) + gradient plane 196x32 fp32 (24.5 KB) + argmax plane (6 KB) are pulled into LDS
  // with one burst of independent loads; the 196 x 25-tap accumulation then never touches global memory.
  float* img = smem;
  float* gl = smem + 1024;
  uint8_t* al = reinterpret_cast<uint8_t*>(smem + 1024 + 7168);
  float* red = smem + 1024 + 7168 + 2048;
  const int tid = threadIdx.x, c = tid & 31, pg = tid >> 5;
  const bool active = tid < 256;
  float acc[26];
#pragma unroll
  for (int t = 0; t < 26; ++t) acc[t] = 0.f;
  for (int b = block; b < B; b += nblocks) {
    float v[4];
    f32x4 ga[4], gb[3];  // ext-vector type: stays in VGPRs (HIP's float4 struct copies land in scratch here)
    uint2 avv[4];
    if (active) {
      const long n = sample_index(perm, step_ptr, spe, B, b);
      const float* src = images + n * 784;
      const f32x4* g4 = reinterpret_cast<const f32x4*>(dpool1 + (long)b * 6272);
      const uint2* a8 = reinterpret_cast<const uint2*>(amax + (long)b * 6272);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, r = e >> 5, cc = e & 31;
        const int y = r - 2, x = cc - 2;
        const bool ok = (unsigned)y < 28u && (unsigned)x < 28u;
        v[i] = src[ok ? y * 28 + x : 0];
        v[i] = ok ? v[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) ga[i] = g4[tid + 256 * i];  // 1568 float4 in total
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int e = tid + 256 * (4 + i);
        gb[i] = g4[e < 1568 ? e : 0];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // 784 uint2
        const int e = tid + 256 * i;
        avv[i] = a8[e < 784 ? e : 0];
      }
    }
    __syncthreads();  // previous image fully consumed
    if (active) {
#pragma unroll
      for (int i = 0; i < 4; ++i) img[tid + 256 * i] = v[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) reinterpret_cast<f32x4*>(gl)[tid + 256 * i] = ga[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) reinterpret_cast<f32x4*>(gl)[tid + 256 * (4 + i)] = gb[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) reinterpret_cast<uint2*>(al)[tid + 256 * i] = avv[i];
    }
    __syncthreads();
    if (active) {
      for (int p = pg; p < 196; p += 8) {
        const float g = gl[p * 32 + c];
        const int a = al[p * 32 + c];
        const int py = p / 14, px = p - py * 14;
        const int r0 = 2 * py + (a >> 1), c0 = 2 * px + (a & 1);
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) acc[dy * 5 + dx] = fmaf(img[(r0 + dy) * 32 + c0 + dx], g, acc[dy * 5 + dx]);
        acc[25] += g;
      }
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < 26; ++t) red[(pg * 26 + t) * 32 + c] = acc[t];
  }
  __syncthreads();
  if (active) {
    for (int e = tid; e < 26 * 32; e += 256) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += red[q * 26 * 32 + e];
      if (e < 25 * 32) atomicAdd(&dW[e], sum); else atomicAdd(&db[e - 25 * 32], sum);
    }
  }
}
