// MNIST-CNN specific kernels (everything that is not GEMM-shaped):
//   conv1 forward  : direct 5x5 conv (Cin=1, K=25 is MFMA-hostile) + bias + ReLU + 2x2 max-pool, with the
//                    mini-batch gathered by index straight out of the HBM-resident dataset
//   conv1 backward : filter/bias gradient through the pool argmax (no data gradient needed)
//   unpool2        : max-pool/ReLU backward that expands the fc1 data gradient into the dense bf16 dY
//                    consumed by the conv2 backward implicit GEMMs, fused with the conv2 bias gradient
//   head           : fc1 split-K combine + bias + ReLU + Philox dropout + fc2 + softmax + (double)
//                    softmax cross-entropy + accuracy + dlogits + fc2/fc1 bias grads + fc1 pre-act grad
// Reference semantics: demo1/train.py:57-141.
#include "common.h"
#include "conv1_dw_body.h"
#include "conv2_pack.h"
#include "conv2_pack3.h"
#include "dtf_structs.h"

namespace {
constexpr int ERR_SHAPE = -22;
constexpr int HID = 1024;
constexpr int NC = 10;
}  // namespace

// ------------------------------------------------------------------ conv1 forward
// one workgroup per (image, group of 8 output channels) (wave-uniform -> weights come through scalar loads).
// x: fp32 [N,784]; Wt: fp32 [25][32] (HWIO, I=1); out: bf16 [B,14,14,32]; amax: uint8 [B,14,14,32].
// F32OUT: the pooled activations are stored as fp32 [B,14,14,32] (reference precision path) instead of bf16.
template <bool F32OUT>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ images, const int* __restrict__ perm,
                                                        const int* __restrict__ step_ptr, int spe, int B,
                                                        const float* __restrict__ Wt, const float* __restrict__ bias,
                                                        void* __restrict__ out_v, uint8_t* __restrict__ amax,
                                                        const float* __restrict__ Wc2, char* __restrict__ Wc2f,
                                                        char* __restrict__ Wc2b) {
  __shared__ __attribute__((aligned(16))) float img[32 * 32];  // rows/cols -2..29 (SAME padding halo)
  __shared__ __attribute__((aligned(16))) float wl[26 * 8];    // this block's 8 filters (25 taps) + bias
  // 1-D grid of npack + 4*B workgroups.  The leading npack ones (dispatched first, so they are no tail) pack conv2's
  // weights for the kernels that follow; the rest are (image, channel group) pairs, channel group fastest.
  const int npack = (int)gridDim.x - 4 * B;
  if ((int)blockIdx.x < npack) {
    if (F32OUT) conv2_pack3_body(Wc2, Wc2f, Wc2b, blockIdx.x, npack);  // (three bf16 planes for the bf16x6 conv2 kernels)
    else conv2_pack_body(Wc2, Wc2f, Wc2b, blockIdx.x, npack);
    return;
  }
  const int b = ((int)blockIdx.x - npack) >> 2, cg = ((int)blockIdx.x - npack) & 3, tid = threadIdx.x;
  const long n = sample_index(perm, step_ptr, spe, B, b);
  const float* src = images + n * 784;
  // every global load of the block is issued before the first wait: one memory round trip
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + 256 * i, r = e >> 5, c = e & 31;
    const int y = r - 2, x = c - 2;
    const bool ok = (unsigned)y < 28u && (unsigned)x < 28u;
    v[i] = src[ok ? y * 28 + x : 0];
    v[i] = ok ? v[i] : 0.f;
  }
  float wv0 = 0.f;
  if (tid < 200) wv0 = Wt[(tid >> 3) * 32 + cg * 8 + (tid & 7)];
  else if (tid < 208) wv0 = bias[cg * 8 + (tid - 200)];
#pragma unroll
  for (int i = 0; i < 4; ++i) img[tid + 256 * i] = v[i];
  if (tid < 208) wl[tid] = wv0;
  __syncthreads();
  if (tid >= 196) return;
  const int py = tid / 14, px = tid - py * 14;
  float patch[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; c += 2) {
      const float2 t = *reinterpret_cast<const float2*>(&img[(2 * py + r) * 32 + 2 * px + c]);
      patch[r][c] = t.x;
      patch[r][c + 1] = t.y;
    }
  // channel pairs on v_pk_fma_f32 (two fp32 fmas per lane and instruction: this loop is the kernel's VALU time)
  f32x2 acc2[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[s][j] = f32x2{0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 5; ++dy)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const float4 w0 = *reinterpret_cast<const float4*>(&wl[(dy * 5 + dx) * 8]);      // LDS broadcast reads
      const float4 w1 = *reinterpret_cast<const float4*>(&wl[(dy * 5 + dx) * 8 + 4]);
      const f32x2 wv[4] = {f32x2{w0.x, w0.y}, f32x2{w0.z, w0.w}, f32x2{w1.x, w1.y}, f32x2{w1.z, w1.w}};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float pv = patch[(s >> 1) + dy][(s & 1) + dx];
        const f32x2 pv2 = f32x2{pv, pv};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[s][j] = __builtin_elementwise_fma(pv2, wv[j], acc2[s][j]);
      }
    }
  float acc[4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[s][2 * j] = acc2[s][j][0];
      acc[s][2 * j + 1] = acc2[s][j][1];
    }
  uint32_t ov[4];
  float of[8];
  uint32_t av[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float best = acc[0][j];
    int arg = 0;
#pragma unroll
    for (int s = 1; s < 4; ++s)
      if (acc[s][j] > best) { best = acc[s][j]; arg = s; }
    const float o = fmaxf(best + wl[200 + j], 0.f);
    of[j] = o;
    const uint32_t bits = f2bf(o);
    if (j & 1) ov[j >> 1] |= bits << 16; else ov[j >> 1] = bits;
    av[j >> 2] |= (uint32_t)arg << ((j & 3) * 8);
  }
  const long o = ((long)b * 196 + tid) * 32 + cg * 8;
  if (F32OUT) {
    float* out = reinterpret_cast<float*>(out_v);
    *reinterpret_cast<f32x4*>(out + o) = f32x4{of[0], of[1], of[2], of[3]};
    *reinterpret_cast<f32x4*>(out + o + 4) = f32x4{of[4], of[5], of[6], of[7]};
  } else {
    uint16_t* out = reinterpret_cast<uint16_t*>(out_v);
    *reinterpret_cast<uint4*>(out + o) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
  }
  if (amax) *reinterpret_cast<uint2*>(amax + o) = make_uint2(av[0], av[1]);
}

// Wc2 != null: also writes the packed bf16 copies of the conv2 filter (conv2_pack.h) into Wc2f / Wc2b.
DTF_API int dtf_conv1_fwd_pack(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                               const float* bias, void* out, void* amax, const float* Wc2, void* Wc2f, void* Wc2b,
                               hipStream_t st) {
  if (B <= 0 || (perm && spe <= 0) || (Wc2 && (!Wc2f || !Wc2b))) return ERR_SHAPE;
  const int npack = Wc2 ? 25 : 0;  // one 16-byte chunk of each packed copy per thread
  hipLaunchKernelGGL(conv1_fwd_kernel<false>, dim3(4 * B + npack), dim3(256), 0, st, images, perm, step_ptr, spe, B, Wt, bias,
                     out, (uint8_t*)amax, Wc2, (char*)Wc2f, (char*)Wc2b);
  DTF_CHECK_LAUNCH();
  return 0;
}

// fp32 pooled activations out [B,14,14,32] (reference-precision path; nothing to pack: the fp32 conv2 kernels read the
// master weights)
DTF_API int dtf_conv1_fwd_f32(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                              const float* bias, float* out, void* amax, hipStream_t st) {
  if (B <= 0 || (perm && spe <= 0)) return ERR_SHAPE;
  hipLaunchKernelGGL(conv1_fwd_kernel<true>, dim3(4 * B), dim3(256), 0, st, images, perm, step_ptr, spe, B, Wt, bias,
                     (void*)out, (uint8_t*)amax, (const float*)nullptr, (char*)nullptr, (char*)nullptr);
  DTF_CHECK_LAUNCH();
  return 0;
}

// the same with the three-plane copies of conv2's filter (conv2_pack3.h) written by 25 leading workgroups
DTF_API int dtf_conv1_fwd_f32_pack3(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                                    const float* bias, float* out, void* amax, const float* Wc2, void* Wf3, void* Wb3,
                                    hipStream_t st) {
  if (B <= 0 || (perm && spe <= 0) || !Wc2 || !Wf3 || !Wb3) return ERR_SHAPE;
  hipLaunchKernelGGL(conv1_fwd_kernel<true>, dim3(4 * B + 25), dim3(256), 0, st, images, perm, step_ptr, spe, B, Wt, bias,
                     (void*)out, (uint8_t*)amax, Wc2, (char*)Wf3, (char*)Wb3);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_conv1_fwd(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                          const float* bias, void* out, void* amax, hipStream_t st) {
  return dtf_conv1_fwd_pack(images, perm, step_ptr, spe, B, Wt, bias, out, amax, nullptr, nullptr, nullptr, st);
}

// ------------------------------------------------------------------ conv1 backward (filter + bias grads)
// dpool1: fp32 [B,14,14,32] gradient wrt the pooled+ReLU'd conv1 output (already ReLU-masked);
// amax: pool argmax.  dW[25][32], db[32] accumulated with atomics (pre-zeroed).
__global__ __launch_bounds__(256) void conv1_dw_kernel(const float* __restrict__ images, const int* __restrict__ perm,
                                                       const int* __restrict__ step_ptr, int spe, int B,
                                                       const float* __restrict__ dpool1, const uint8_t* __restrict__ amax,
                                                       float* __restrict__ dW, float* __restrict__ db) {
  __shared__ __attribute__((aligned(16))) float smem[CONV1_DW_LDS_FLOATS];
  conv1_dw_body(smem, blockIdx.x, gridDim.x, images, perm, step_ptr, spe, B, dpool1, amax, dW, db);
}

DTF_API int dtf_conv1_dw(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* dpool1,
                         const void* amax, float* dW, float* db, hipStream_t st) {
  if (B <= 0 || (perm && spe <= 0)) return ERR_SHAPE;
  const int grid = B < 256 ? B : 256;
  hipLaunchKernelGGL(conv1_dw_kernel, dim3(grid), dim3(256), 0, st, images, perm, step_ptr, spe, B, dpool1,
                     (const uint8_t*)amax, dW, db);
  DTF_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ unpool2 (+ conv2 bias grad)
// dpool: fp32 [W][C] (W = B*Hp*Wp windows), pooled: bf16 [W][C] forward output (ReLU mask), amax: uint8 [W][C].
// dY: bf16 dense [B, 2Hp, 2Wp, C]; db[C] += sum of the masked gradient.   C % 8 == 0, C <= 256.
__global__ __launch_bounds__(256) void unpool_kernel(const float* __restrict__ dpool, const uint16_t* __restrict__ pooled,
                                                     const uint8_t* __restrict__ amax, uint16_t* __restrict__ dY,
                                                     float* __restrict__ db, int nwin, int Hp, int Wp, int C) {
  __shared__ float dbs[256];
  const int tid = threadIdx.x;
  dbs[tid] = 0.f;
  __syncthreads();
  const int c8n = C >> 3;
  const long gi = (long)blockIdx.x * 256 + tid;
  const long win = gi / c8n;
  const int c0 = (int)(gi - win * c8n) * 8;
  if (win < nwin) {
    const long o = win * C + c0;
    const float4 g0 = *reinterpret_cast<const float4*>(dpool + o);
    const float4 g1 = *reinterpret_cast<const float4*>(dpool + o + 4);
    const uint4 pv = *reinterpret_cast<const uint4*>(pooled + o);
    const uint2 av = *reinterpret_cast<const uint2*>(amax + o);
    float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
    int a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t bits = (pw[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
      const bool pos = (bits & 0x7fffu) != 0 && (bits & 0x8000u) == 0;
      g[j] = pos ? g[j] : 0.f;
      a[j] = ((j < 4 ? av.x : av.y) >> ((j & 3) * 8)) & 0xff;
      if (db) atomicAdd(&dbs[c0 + j], g[j]);
    }
    const int px = (int)(win % Wp);
    const long t = win / Wp;
    const int py = (int)(t % Hp);
    const long b = t / Hp;
    const int W2 = 2 * Wp;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint32_t ov[4];
#pragma unroll
      for (int j = 0; j < 8; j += 2)
        ov[j >> 1] = pack_bf2(a[j] == s ? g[j] : 0.f, a[j + 1] == s ? g[j + 1] : 0.f);
      const long d = ((b * 2 * Hp + 2 * py + (s >> 1)) * W2 + 2 * px + (s & 1)) * C + c0;
      *reinterpret_cast<uint4*>(dY + d) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
    }
  }
  if (db) {
    __syncthreads();
    if (tid < C) atomicAdd(&db[tid], dbs[tid]);
  }
}

DTF_API int dtf_unpool(const float* dpool, const void* pooled, const void* amax, void* dY, float* db, int B, int Hp,
                       int Wp, int C, hipStream_t st) {
  if (B <= 0 || C % 8 || C > 256) return ERR_SHAPE;
  const long nwin = (long)B * Hp * Wp;
  const long nthreads = nwin * (C / 8);
  hipLaunchKernelGGL(unpool_kernel, dim3(dtf_cdiv(nthreads, 256)), dim3(256), 0, st, dpool, (const uint16_t*)pooled,
                     (const uint8_t*)amax, (uint16_t*)dY, db, (int)nwin, Hp, Wp, C);
  DTF_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ head

// (fast hardware exp/log on the bf16 path, the accurate library forms on the reference-precision path)
template <bool F32> __device__ __forceinline__ float head_exp(float x) { return F32 ? expf(x) : __expf(x); }
template <bool F32> __device__ __forceinline__ float head_log(float x) { return F32 ? logf(x) : __logf(x); }

// F32: h_drop / dh / dz are stored as fp32 (dz rows padded to 16 floats) and nothing is rounded to bf16.
template <bool F32>
__global__ __launch_bounds__(256) void head_kernel(const HeadParams p) {
  __shared__ float red[2][4][NC];
  __shared__ float db1s[HID];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = tid * 4;
  float w2[4][NC];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < NC; ++j) w2[q][j] = p.W_fc2[(k + q) * NC + j];
  const float4 b1 = *reinterpret_cast<const float4*>(p.b_fc1 + k);
  float b2[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) b2[j] = p.b_fc2[j];
  const int step = p.step_ptr ? *p.step_ptr : 0;
  if (p.zero_buf) {
    for (long i = (long)blockIdx.x * 256 + tid; i < p.zero_n4; i += (long)gridDim.x * 256)
      reinterpret_cast<float4*>(p.zero_buf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float accb1[4] = {0.f, 0.f, 0.f, 0.f};
  float accb2[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) accb2[j] = 0.f;
  float loss_sum = 0.f, correct = 0.f;
  const float inv_keep = 1.f / p.keep_prob;
  int it = 0;
  for (int row = blockIdx.x; row < p.B; row += gridDim.x, ++it) {
    float4 h = b1;
    // the row's label: two dependent loads (permutation, label) that would otherwise start behind the barrier below
    const int y = p.labels_all[sample_index(p.perm, p.step_ptr, p.spe, p.B, row)];
    // all split-K slabs of this row in flight at once (clamped index + select: one L2 round trip instead of nslab/4)
    for (int s0 = 0; s0 < p.nslab; s0 += 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int s = s0 + u < p.nslab ? s0 + u : p.nslab - 1;
        v[u] = *reinterpret_cast<const f32x4*>(p.slabs + (long)s * p.slab_stride + (long)row * HID + k);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (s0 + u < p.nslab) { h.x += v[u][0]; h.y += v[u][1]; h.z += v[u][2]; h.w += v[u][3]; }
    }
    float hv[4] = {fmaxf(h.x, 0.f), fmaxf(h.y, 0.f), fmaxf(h.z, 0.f), fmaxf(h.w, 0.f)};
    bool keep[4] = {true, true, true, true};
    if (p.training && p.keep_prob < 1.f) {
      const uint4 r = philox4x32_10(make_uint4((uint32_t)(row * 256 + tid), (uint32_t)step, 0x6d6e6973u, 0u),
                                    make_uint2((uint32_t)p.seed, (uint32_t)(p.seed >> 32)));
      const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        keep[q] = u32_to_unit(rv[q]) < p.keep_prob;
        hv[q] = keep[q] ? hv[q] * inv_keep : 0.f;
      }
    }
    // the fc2 GEMV consumes the same bf16-rounded activations the backward GEMMs will see
    uint32_t hb[2] = {pack_bf2(hv[0], hv[1]), pack_bf2(hv[2], hv[3])};
    float hr[4] = {bf2f(hb[0] & 0xffffu), bf2f(hb[0] >> 16), bf2f(hb[1] & 0xffffu), bf2f(hb[1] >> 16)};
    if (F32) {
#pragma unroll
      for (int q = 0; q < 4; ++q) hr[q] = hv[q];
    }
    // The wave's NC sums over 64 lanes as ONE transposing reduction (13 lane exchanges instead of NC x 6): at the step
    // with partner lane ^ m, a lane keeps the classes whose index has the lane's bit m and hands the others over; after
    // m = 1, 2, 4, 8 lane l holds class (l & 15) summed over its 16 lanes, m = 16, 32 finish it.
    float v8[8], v4[4], v2[2];
    {
      float part[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fmaf(hr[q], w2[q][j], s);
        part[j] = s;
      }
      static_assert(NC == 10, "the exchange pattern below is written for 10 classes padded to 16");
      const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
      for (int i = 0; i < 5; ++i)  // classes 2i, 2i + 1
        v8[i] = (b0 ? part[2 * i + 1] : part[2 * i]) + __shfl_xor(b0 ? part[2 * i] : part[2 * i + 1], 1);
      v4[0] = (b1 ? v8[1] : v8[0]) + __shfl_xor(b1 ? v8[0] : v8[1], 2);
      v4[1] = (b1 ? v8[3] : v8[2]) + __shfl_xor(b1 ? v8[2] : v8[3], 2);
      v4[2] = (b1 ? 0.f : v8[4]) + __shfl_xor(b1 ? v8[4] : 0.f, 2);   // (classes 10, 11 are padding)
      v2[0] = (b2 ? v4[1] : v4[0]) + __shfl_xor(b2 ? v4[0] : v4[1], 4);
      v2[1] = (b2 ? 0.f : v4[2]) + __shfl_xor(b2 ? v4[2] : 0.f, 4);   // (classes 12 .. 15 are padding)
      float t = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      v8[0] = t;  // class (lane & 15) of this wave
    }
    const int buf = it & 1;
    if (lane < NC) red[buf][wave][lane] = v8[0];
    __syncthreads();
    float z[NC], zmax = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      z[j] = red[buf][0][j] + red[buf][1][j] + red[buf][2][j] + red[buf][3][j] + b2[j];
      zmax = fmaxf(zmax, z[j]);
    }
    float pr[NC], psum = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) { pr[j] = head_exp<F32>(z[j] - zmax); psum += pr[j]; }
    const float pinv = 1.f / psum;
    int arg = 0;
    float pbest = -1.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      pr[j] *= pinv;
      if (pr[j] > pbest) { pbest = pr[j]; arg = j; }
    }
    float dzv[NC];
    float loss;
    if (p.double_softmax) {
      // loss = -log softmax(p)[y]   (demo1/train.py:123,127: softmax applied twice)
      float q[NC], qsum = 0.f;
#pragma unroll
      for (int j = 0; j < NC; ++j) { q[j] = head_exp<F32>(pr[j] - 1.f); qsum += q[j]; }  // p <= 1: stable shift
      const float qinv = 1.f / qsum;
      float gp[NC], dot = 0.f, qy = 0.f;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        q[j] *= qinv;
        if (j == y) qy = q[j];
        gp[j] = (q[j] - (j == y ? 1.f : 0.f)) * p.grad_scale;
        dot = fmaf(gp[j], pr[j], dot);
      }
      loss = -head_log<F32>(qy);
#pragma unroll
      for (int j = 0; j < NC; ++j) dzv[j] = pr[j] * (gp[j] - dot);
    } else {
      float zy = 0.f;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        if (j == y) zy = z[j];
        dzv[j] = (pr[j] - (j == y ? 1.f : 0.f)) * p.grad_scale;
      }
      loss = head_log<F32>(psum) - (zy - zmax);  // log-sum-exp form: exact even when p[y] underflows
    }
    loss_sum += loss;  // (every thread holds the row's logits: no broadcast needed for the block's totals)
    correct += (arg == y) ? 1.f : 0.f;
    if (tid == 0 && p.pred) p.pred[row] = arg;
    if (p.probs) {
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (tid == j) p.probs[(long)row * NC + j] = pr[j];
    }
    if (p.training) {
      if (F32)
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.h_drop) + (long)row * HID + k) = f32x4{hv[0], hv[1], hv[2], hv[3]};
      else
        *reinterpret_cast<uint2*>(p.h_drop + (long)row * HID + k) = make_uint2(hb[0], hb[1]);
      float dhv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j) s = fmaf(dzv[j], w2[q][j], s);
        dhv[q] = (keep[q] && hv[q] > 0.f) ? s * inv_keep : 0.f;
        accb1[q] += dhv[q];
      }
      if (F32)
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dh) + (long)row * HID + k) = f32x4{dhv[0], dhv[1], dhv[2], dhv[3]};
      else
        *reinterpret_cast<uint2*>(p.dh + (long)row * HID + k) = make_uint2(pack_bf2(dhv[0], dhv[1]), pack_bf2(dhv[2], dhv[3]));
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (tid == j) {
          if (F32) reinterpret_cast<float*>(p.dz)[(long)row * 16 + j] = dzv[j];
          else p.dz[(long)row * NC + j] = f2bf(dzv[j]);
        }
      // fp32 with the bias gradients taken by the fused fc backward launch: that launch also adds up the rows' loss and
      // hit count, left in two padding columns of dz (no operand chunk of the fc2 filter gradient reaches columns 12..15)
      if (F32 && !p.db_fc1 && (tid == 12 || tid == 13))
        reinterpret_cast<float*>(p.dz)[(long)row * 16 + tid] = tid == 12 ? loss : ((arg == y) ? 1.f : 0.f);
#pragma unroll
      for (int j = 0; j < NC; ++j) accb2[j] += dzv[j];
    }
  }
  // One atomic *instruction* per destination line: every workgroup adds to the same two cache lines, and the L2
  // serialises same-line requests (12 single-lane atomics per workgroup cost ~3 us of a 17 us kernel at 50 workgroups).
  if (tid < 2 && !(F32 && p.training && !p.db_fc1)) {
    float* m = p.metrics + (p.metrics_ring > 0 ? 2 * (step % p.metrics_ring) : 0);
    atomicAdd(&m[tid], tid == 0 ? loss_sum : correct);
  }
  if (p.training && p.db_fc1) {  // (null: the fused fc backward launch takes the bias gradients as column sums)
    float mine = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) mine = tid == j ? accb2[j] : mine;
    if (tid < NC) atomicAdd(&p.db_fc2[tid], mine);
    // contiguous 256-B atomics per wave instruction (lane-strided float4 adds are ~4x more memory requests)
#pragma unroll
    for (int q = 0; q < 4; ++q) db1s[k + q] = accb1[q];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicAdd(&p.db_fc1[i * 256 + tid], db1s[i * 256 + tid]);
  }
}

DTF_API int dtf_head(const HeadParams* hp, int blocks, hipStream_t st) {
  if (hp->B <= 0 || hp->nslab < 1 || blocks < 1 || hp->keep_prob <= 0.f) return ERR_SHAPE;
  if (hp->training && (!hp->h_drop || !hp->dh || !hp->dz || (hp->db_fc1 == nullptr) != (hp->db_fc2 == nullptr))) return ERR_SHAPE;
  if (blocks > hp->B) blocks = hp->B;
  if (hp->f32)
    hipLaunchKernelGGL(head_kernel<true>, dim3(blocks), dim3(256), 0, st, *hp);
  else
    hipLaunchKernelGGL(head_kernel<false>, dim3(blocks), dim3(256), 0, st, *hp);
  DTF_CHECK_LAUNCH();
  return 0;
}
