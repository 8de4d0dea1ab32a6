// Training-mode kernels for the Inception-v3 full fine-tune path (BASELINE config 5; SURVEY §2.6 last paragraph):
//   * conv forward writing the raw bf16 pre-activation and accumulating the batch-norm statistics in its epilogue,
//   * batch-norm finalize / normalise+ReLU forward / two-pass backward (column reduce, then apply in place),
//   * conv backward-data for every geometry of the network (1x1, 3x3, 5x5, 1x7, 7x1, 1x3, 3x1; stride 1/2; SAME/VALID),
//   * conv filter gradient for the same geometries,
//   * max-pool 3x3/2 (argmax kept), avg-pool 3x3 SAME (TF divisor) and global average pool backward.
// Master weights are fp32 HWIO ([KH*KW*Cin][Cout], TensorFlow's layout) and are converted to bf16 while staged into
// LDS, so the optimizer updates one fp32 arena and no shadow copy is maintained.  Activations and activation
// gradients are bf16 NHWC; a tensor may be a channel slice of a wider concat buffer (explicit channel strides).
#include "gemm_core.h"
#include "dtf_api.h"

namespace {
constexpr int ERR_SHAPE = -22;

inline GemmParams mk_params(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc,
                            int splits, int BK) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  const int ksteps = dtf_cdiv(K, BK);
  p.kchunk = dtf_cdiv(ksteps, splits < 1 ? 1 : splits) * BK;
  p.slab_stride = (long)M * ldc;
  return p;
}
inline int nsplits(const GemmParams& p) { return dtf_cdiv(p.K, p.kchunk); }

inline int geom(GemmParams& p, int H, int W, int C, int cstride, int Ho, int Wo, int KH, int KW, int stride, int pad_h,
                int pad_w) {
  p.g.H = H; p.g.W = W; p.g.Cin = C; p.g.cstride = cstride; p.g.Ho = Ho; p.g.Wo = Wo;
  p.g.KH = KH; p.g.KW = KW; p.g.stride = stride; p.g.pad_h = pad_h; p.g.pad_w = pad_w;
  if (C % 8 || cstride % 8 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return ERR_SHAPE;
  if ((H + 2 * pad_h - KH) / stride + 1 != Ho || (W + 2 * pad_w - KW) / stride + 1 != Wo) return ERR_SHAPE;
  return 0;
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = bf2f((uint16_t)(w[j] & 0xffffu));
    f[2 * j + 1] = bf2f((uint16_t)(w[j] >> 16));
  }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}
__device__ __forceinline__ void ld8f(const float* p, float* f) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
inline int ew_blocks(long total, int cap = 8192) {
  long b = (total + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}
}  // namespace

// ------------------------------------------------------------------------------------------------ conv forward
// z[B,Ho,Wo,Cout] (bf16, dense) = conv(x, Wt);  stats[2][Cout] += (sum z, sum z^2) per output channel.
DTF_API int dtf_conv_fwd_stats(const void* x, const float* Wt, void* z, float* stats, int B, int H, int W, int Cin,
                               int in_cstride, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int Ho,
                               int Wo, hipStream_t st) {
  if (B <= 0 || Cout <= 0 || Cout % 2) return ERR_SHAPE;
  const int M = B * Ho * Wo, K = KH * KW * Cin;
  GemmParams p = mk_params(x, Wt, z, M, Cout, K, 0, Cout, Cout, 1, 64);
  if (geom(p, H, W, Cin, in_cstride, Ho, Wo, KH, KW, stride, pad_h, pad_w)) return ERR_SHAPE;
  p.aux_out = stats;
  if (Cout <= 32)
    return dtf_launch_gemm<128, 32, 64, 4, 1, LoadConvA<128, 256, 64, 0>, LoadColK<float, 32, 256, 64, false>, EpBf16Stats>(p, 1, st);
  if (M <= 256 * 64)
    return dtf_launch_gemm<64, 64, 64, 2, 2, LoadConvA<64, 256, 64, 0>, LoadColK<float, 64, 256, 64, false>, EpBf16Stats>(p, 1, st);
  return dtf_launch_gemm<128, 64, 64, 4, 1, LoadConvA<128, 256, 64, 0>, LoadColK<float, 64, 256, 64, false>, EpBf16Stats>(p, 1, st);
}

// ------------------------------------------------------------------------------------------------ conv backward-data
// dx[B,H,W,Cin] (bf16 at channel stride dx_cstride; overwritten or accumulated) = conv^T(dz, Wt)
DTF_API int dtf_conv_bwd_data(const void* dz, const float* Wt, void* dx, int B, int H, int W, int Cin, int dx_cstride,
                              int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int Ho, int Wo, int accumulate,
                              hipStream_t st) {
  if (B <= 0 || Cin <= 0 || Cin % 8 || (stride != 1 && stride != 2)) return ERR_SHAPE;
  const int M = B * H * W, K = KH * KW * Cout;
  GemmParams p = mk_params(dz, Wt, dx, M, Cin, K, 0, 0, dx_cstride, 1, 64);
  if (geom(p, H, W, Cout, Cout, Ho, Wo, KH, KW, stride, pad_h, pad_w)) return ERR_SHAPE;
  p.accumulate = accumulate;
  if (M <= 256 * 64)
    return dtf_launch_gemm<64, 64, 64, 2, 2, LoadConvBwdA<64, 256, 64>, LoadConvWBwd<64, 256, 64, false>, EpBf16Acc>(p, 1, st);
  return dtf_launch_gemm<128, 64, 64, 4, 1, LoadConvBwdA<128, 256, 64>, LoadConvWBwd<64, 256, 64, false>, EpBf16Acc>(p, 1, st);
}

// ------------------------------------------------------------------------------------------------ conv filter gradient
// dW[(tap,cin)][cout] (fp32 HWIO, pre-zeroed; split-K atomics) += sum_pos x[pos (+) tap][cin] * dz[pos][cout]
DTF_API int dtf_conv_dw_general(const void* x, int in_cstride, const void* dz, float* dW, int B, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int Ho, int Wo, int splits,
                                hipStream_t st) {
  if (B <= 0 || Cin % 2 || Cout % 2 || splits < 1) return ERR_SHAPE;
  const int M = KH * KW * Cin, K = B * Ho * Wo;
  GemmParams p = mk_params(x, dz, dW, M, Cout, K, 0, Cout, Cout, splits, 64);
  if (geom(p, H, W, Cin, in_cstride, Ho, Wo, KH, KW, stride, pad_h, pad_w)) return ERR_SHAPE;
  return dtf_launch_gemm<128, 64, 64, 4, 1, LoadConvDwA<128, 256, 64>, LoadColK<uint16_t, 64, 256, 64, false>, EpAtomicF32>(
      p, nsplits(p), st);
}

// ------------------------------------------------------------------------------------------------ batch norm
// bn[4][C] = mean, rstd, scale (= gamma*rstd), shift (= beta - mean*scale) from stats[DTF_STAT_SLOTS][2][C] (which is
// cleared again); moving stats updated in place.
__global__ __launch_bounds__(256) void bn_finalize_kernel(float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ mmean,
                                                          float* __restrict__ mvar, float* __restrict__ bn, int C,
                                                          float inv_count, float eps, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < DTF_STAT_SLOTS; ++k) {
    s += stats[(2 * k) * C + c];
    ss += stats[(2 * k + 1) * C + c];
    stats[(2 * k) * C + c] = 0.f;  // re-arm the accumulators for the next step (this kernel is their only reader)
    stats[(2 * k + 1) * C + c] = 0.f;
  }
  const float mean = s * inv_count;
  const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  const float scale = gamma[c] * rstd;
  bn[c] = mean;
  bn[C + c] = rstd;
  bn[2 * C + c] = scale;
  bn[3 * C + c] = beta[c] - mean * scale;
  if (mmean) {
    mmean[c] = momentum * mmean[c] + (1.f - momentum) * mean;
    mvar[c] = momentum * mvar[c] + (1.f - momentum) * var;
  }
}

DTF_API int dtf_bn_finalize(float* stats, const float* gamma, const float* beta, float* moving_mean,
                            float* moving_var, float* bn, int C, long count, float eps, float momentum, hipStream_t st) {
  if (C <= 0 || count <= 0) return ERR_SHAPE;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(dtf_cdiv(C, 256)), dim3(256), 0, st, stats, gamma, beta, moving_mean,
                     moving_var, bn, C, 1.f / (float)count, eps, momentum);
  DTF_CHECK_LAUNCH();
  return 0;
}

// y[m*y_cs + c] = bf16(relu(z[m][c]*scale[c] + shift[c]))
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const uint16_t* __restrict__ z, const float* __restrict__ bn,
                                                          uint16_t* __restrict__ y, long M, int C, int y_cs) {
  const int c8n = C >> 3;
  const long total = M * c8n;
  for (long gi = (long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long)gridDim.x * 256) {
    const int c0 = (int)(gi % c8n) * 8;
    const long m = gi / c8n;
    float f[8], sc[8], sh[8];
    unpack8(*reinterpret_cast<const uint4*>(z + m * C + c0), f);
    ld8f(bn + 2 * C + c0, sc);
    ld8f(bn + 3 * C + c0, sh);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
    *reinterpret_cast<uint4*>(y + m * y_cs + c0) = pack8(f);
  }
}

DTF_API int dtf_bn_relu_fwd(const void* z, const float* bn, void* y, long M, int C, int y_cstride, hipStream_t st) {
  if (M <= 0 || C <= 0 || C % 8 || y_cstride % 8) return ERR_SHAPE;
  hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(ew_blocks(M * (C / 8))), dim3(256), 0, st, (const uint16_t*)z, bn,
                     (uint16_t*)y, M, C, y_cstride);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Finalize + normalise + ReLU in one launch (saves the 94 one-workgroup finalize launches of a step): every workgroup
// folds the DTF_STAT_SLOTS statistic replicas into LDS scale/shift itself (2C x 16 L2-resident floats); workgroup 0
// also stores bn[4][C] for the backward, updates the moving statistics and clears `stats_prev`, the replica set of the
// previous step (the two sets alternate, so no workgroup of this launch reads it).
__global__ __launch_bounds__(256) void bn_finalize_relu_fwd_kernel(const float* __restrict__ stats, float* __restrict__ stats_prev,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ mmean, float* __restrict__ mvar,
                                                                   float* __restrict__ bn, const uint16_t* __restrict__ z,
                                                                   uint16_t* __restrict__ y, long M, int C, int y_cs,
                                                                   float inv_count, float eps, float momentum) {
  extern __shared__ float sm[];  // scale[C], shift[C]
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int k = 0; k < DTF_STAT_SLOTS; ++k) {
      s += stats[(2 * k) * C + c];
      ss += stats[(2 * k + 1) * C + c];
    }
    const float mean = s * inv_count;
    const float var = fmaxf(ss * inv_count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float scale = gamma[c] * rstd, shift = beta[c] - mean * scale;
    sm[c] = scale;
    sm[C + c] = shift;
    if (blockIdx.x == 0) {
      bn[c] = mean;
      bn[C + c] = rstd;
      bn[2 * C + c] = scale;
      bn[3 * C + c] = shift;
      if (mmean) {
        mmean[c] = momentum * mmean[c] + (1.f - momentum) * mean;
        mvar[c] = momentum * mvar[c] + (1.f - momentum) * var;
      }
      if (stats_prev)
#pragma unroll
        for (int k = 0; k < 2 * DTF_STAT_SLOTS; ++k) stats_prev[k * C + c] = 0.f;
    }
  }
  __syncthreads();
  // A lane keeps ONE group of 8 channels (scale / shift in registers, no per-element LDS or parameter reads) and walks the
  // rows with four 16-byte loads in flight (clamped address + select): a one-load-per-iteration grid-stride loop keeps
  // ~4 MB in flight on the whole chip, less than HBM latency x bandwidth needs.
  const int c8n = C >> 3, rpb = 256 / c8n;
  const int ry = threadIdx.x / c8n, c0 = (threadIdx.x - ry * c8n) * 8;
  if (ry >= rpb) return;
  float sc[8], sf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = sm[c0 + j]; sf[j] = sm[C + c0 + j]; }
  const long stride = (long)gridDim.x * rpb;
  for (long r = (long)blockIdx.x * rpb + ry; r < M; r += 4 * stride) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = r + u * stride < M ? r + u * stride : r;
      v[u] = *reinterpret_cast<const uint4*>(z + rr * C + c0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sf[j], 0.f);
      if (r + u * stride < M) *reinterpret_cast<uint4*>(y + (r + u * stride) * y_cs + c0) = pack8(f);
    }
  }
}

DTF_API int dtf_bn_finalize_relu_fwd(const float* stats, float* stats_prev, const float* gamma, const float* beta,
                                     float* moving_mean, float* moving_var, float* bn, const void* z, void* y, long M, int C,
                                     int y_cstride, float eps, float momentum, hipStream_t st) {
  if (M <= 0 || C <= 0 || C % 8 || C > 2048 || y_cstride % 8) return ERR_SHAPE;
  const int blocks = ew_blocks(M * (C / 8) / 4, 1024);  // >= 4 elements-of-8 per lane amortise the fold
  hipLaunchKernelGGL(bn_finalize_relu_fwd_kernel, dim3(blocks), dim3(256), 2 * C * sizeof(float), st, stats, stats_prev, gamma,
                     beta, moving_mean, moving_var, bn, (const uint16_t*)z, (uint16_t*)y, M, C, y_cstride, 1.f / (float)M, eps,
                     momentum);
  DTF_CHECK_LAUNCH();
  return 0;
}

// slots[DTF_STAT_SLOTS][2][C] += (sum g, sum g*xhat) with g = dy * (y > 0), xhat = (z - mean) * rstd.
// The ReLU mask is recomputed from z (y > 0  <=>  z*scale + shift > 0) instead of re-reading the stored activation:
// one third less traffic in both backward passes.
// Workgroups accumulate into DTF_STAT_SLOTS replicas (atomic contention, see EpBf16Stats).  No fence / ticket here:
// an agent-scope release on this multi-XCD part writes back the whole L2 (measured ~18 us per launch); the apply
// kernel that follows folds the replicas itself.
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const uint16_t* __restrict__ dy, int dy_cs,
                                                                 const uint16_t* __restrict__ z, const float* __restrict__ bn,
                                                                 float* __restrict__ slots, long M, int C, int rows_per_block) {
  __shared__ float red[256 * 16];
  const int c8n = C >> 3;
  const int rpb = 256 / c8n;
  const int tid = threadIdx.x;
  const int ry = tid / c8n, cg = tid - ry * c8n, c0 = cg * 8;
  const bool active = ry < rpb;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (active) {
    float mean[8], rstd[8], sc[8], sf[8];
    ld8f(bn + c0, mean);
    ld8f(bn + C + c0, rstd);
    ld8f(bn + 2 * C + c0, sc);
    ld8f(bn + 3 * C + c0, sf);
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (long r = r0 + ry; r < r1; r += 4L * rpb) {  // four rows in flight per lane (clamped address + select)
      uint4 gv[4], zv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long rr = r + (long)u * rpb < r1 ? r + (long)u * rpb : r;
        gv[u] = *reinterpret_cast<const uint4*>(dy + rr * dy_cs + c0);
        zv[u] = *reinterpret_cast<const uint4*>(z + rr * C + c0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = r + (long)u * rpb < r1;
        float g[8], zz[8];
        unpack8(gv[u], g); unpack8(zv[u], zz);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gj = (ok && zz[j] * sc[j] + sf[j] > 0.f) ? g[j] : 0.f;
          s1[j] += gj;
          s2[j] += gj * (zz[j] - mean[j]) * rstd[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[tid * 16 + j] = s1[j];
    red[tid * 16 + 8 + j] = s2[j];
  }
  __syncthreads();
  if (active && ry == 0) {
    for (int k = 1; k < rpb; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s1[j] += red[(k * c8n + cg) * 16 + j];
        s2[j] += red[(k * c8n + cg) * 16 + 8 + j];
      }
  }
  // channel-contiguous atomics: a lane-strided add (lane = 8 channels) sends every destination line 8 requests per
  // workgroup, and the L2 serialises requests to one line; through LDS each line gets one
  __syncthreads();
  if (active && ry == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[c0 + j] = s1[j];
      red[C + c0 + j] = s2[j];
    }
  }
  __syncthreads();
  float* slot = slots + (long)(blockIdx.x % DTF_STAT_SLOTS) * 2 * C;
  for (int e = tid; e < 2 * C; e += 256) atomicAdd(&slot[e], red[e]);
}

DTF_API int dtf_bn_relu_bwd_reduce(const void* dy, int dy_cstride, const void* z, const float* bn, float* slots, long M,
                                   int C, hipStream_t st) {
  if (M <= 0 || C <= 0 || C % 8 || C > 2048 || dy_cstride % 8) return ERR_SHAPE;
  const int rpb = 256 / (C / 8);
  long rows = (M + 1023) / 1024;
  if (rows < 4L * rpb) rows = 4L * rpb;
  hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(dtf_cdiv(M, rows)), dim3(256), 0, st, (const uint16_t*)dy, dy_cstride,
                     (const uint16_t*)z, bn, slots, M, C, (int)rows);
  DTF_CHECK_LAUNCH();
  return 0;
}

// z <- dz = scale * (g - s1/M - xhat * s2/M) in place.  Every workgroup first folds the DTF_STAT_SLOTS replicas of
// (s1, s2) into LDS (2C x 16 L2-resident floats); workgroup 0 also emits dbeta = s1, dgamma = s2 and clears
// `slots_prev` — the replica set the previous step used (the two sets alternate, so nobody is reading it any more).
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const uint16_t* __restrict__ dy, int dy_cs,
                                                                uint16_t* __restrict__ z, const float* __restrict__ bn,
                                                                const float* __restrict__ slots, float* __restrict__ slots_prev,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, long M,
                                                                int C, float inv_m) {
  extern __shared__ float sm[];  // [2C]
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < DTF_STAT_SLOTS; ++k) t += slots[(long)k * 2 * C + i];
    sm[i] = t;
    if (blockIdx.x == 0) {
      if (dgamma) {
        if (i < C) dbeta[i] = t;
        else dgamma[i - C] = t;
      }
      if (slots_prev)
#pragma unroll
        for (int k = 0; k < DTF_STAT_SLOTS; ++k) slots_prev[(long)k * 2 * C + i] = 0.f;
    }
  }
  __syncthreads();
  // one group of 8 channels per lane, parameters in registers, four rows in flight (see bn_finalize_relu_fwd_kernel)
  const int c8n = C >> 3, rpb = 256 / c8n;
  const int ry = threadIdx.x / c8n, c0 = (threadIdx.x - ry * c8n) * 8;
  if (ry >= rpb) return;
  float mean[8], rstd[8], sc[8], sf[8], a1[8], a2[8];
  ld8f(bn + c0, mean);
  ld8f(bn + C + c0, rstd);
  ld8f(bn + 2 * C + c0, sc);
  ld8f(bn + 3 * C + c0, sf);
#pragma unroll
  for (int j = 0; j < 8; ++j) { a1[j] = sm[c0 + j] * inv_m; a2[j] = sm[C + c0 + j] * inv_m; }
  const long stride = (long)gridDim.x * rpb;
  for (long r = (long)blockIdx.x * rpb + ry; r < M; r += 4 * stride) {
    uint4 gv[4], zv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = r + u * stride < M ? r + u * stride : r;
      gv[u] = *reinterpret_cast<const uint4*>(dy + rr * dy_cs + c0);
      zv[u] = *reinterpret_cast<const uint4*>(z + rr * C + c0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float g[8], zz[8];
      unpack8(gv[u], g);
      unpack8(zv[u], zz);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gj = zz[j] * sc[j] + sf[j] > 0.f ? g[j] : 0.f;
        const float xh = (zz[j] - mean[j]) * rstd[j];
        zz[j] = sc[j] * (gj - a1[j] - xh * a2[j]);
      }
      if (r + u * stride < M) *reinterpret_cast<uint4*>(z + (r + u * stride) * C + c0) = pack8(zz);
    }
  }
}

DTF_API int dtf_bn_relu_bwd_apply(const void* dy, int dy_cstride, void* z, const float* bn, const float* slots,
                                  float* slots_prev, float* dgamma, float* dbeta, long M, int C, hipStream_t st) {
  if (M <= 0 || C <= 0 || C % 8 || C > 2048 || dy_cstride % 8) return ERR_SHAPE;
  int blocks = ew_blocks(M * (C / 8) / 4, 2048);  // >= 4 elements-of-8 per lane amortise the fold
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(blocks), dim3(256), 2 * C * sizeof(float), st, (const uint16_t*)dy,
                     dy_cstride, (uint16_t*)z, bn, slots, slots_prev, dgamma, dbeta, M, C, 1.f / (float)M);
  DTF_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------ pools
// max 3x3 stride s VALID, keeping the window-relative argmax (0..8, first maximum like TF's MaxPoolGrad)
__global__ __launch_bounds__(256) void maxpool3x3_arg_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                             uint8_t* __restrict__ arg, int B, int H, int W, int C, int in_cs,
                                                             int Ho, int Wo, int out_cs, int stride) {
  const int c8n = C >> 3;
  const long total = (long)B * Ho * Wo * c8n;
  for (long gi = (long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long)gridDim.x * 256) {
    const int c0 = (int)(gi % c8n) * 8;
    long t = gi / c8n;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    uint4 v[9];  // (all nine taps requested before the first compare)
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
      v[t9] = *reinterpret_cast<const uint4*>(x + (((long)b * H + ho * stride + t9 / 3) * W + wo * stride + t9 % 3) * in_cs + c0);
    float best[8];
    unsigned am[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -3.0e38f; am[j] = 0; }
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      float f[8];
      unpack8(v[t9], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool gt = f[j] > best[j];
        best[j] = gt ? f[j] : best[j];
        am[j] = gt ? (unsigned)t9 : am[j];
      }
    }
    const long o = ((long)b * Ho + ho) * Wo + wo;
    *reinterpret_cast<uint4*>(out + o * out_cs + c0) = pack8(best);
    *reinterpret_cast<uint2*>(arg + o * C + c0) =
        make_uint2(am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24), am[4] | (am[5] << 8) | (am[6] << 16) | (am[7] << 24));
  }
}

DTF_API int dtf_maxpool3x3_arg(const void* x, void* out, void* argmax, int B, int H, int W, int C, int in_cs, int out_cs,
                               int stride, hipStream_t st) {
  if (B <= 0 || C % 8 || in_cs % 8 || out_cs % 8 || H < 3 || W < 3 || stride < 1) return ERR_SHAPE;
  const int Ho = (H - 3) / stride + 1, Wo = (W - 3) / stride + 1;
  hipLaunchKernelGGL(maxpool3x3_arg_kernel, dim3(ew_blocks((long)B * Ho * Wo * (C / 8))), dim3(256), 0, st,
                     (const uint16_t*)x, (uint16_t*)out, (uint8_t*)argmax, B, H, W, C, in_cs, Ho, Wo, out_cs, stride);
  DTF_CHECK_LAUNCH();
  return 0;
}

// dx[b,h,w,c] (+)= sum over the windows (ho,wo) containing (h,w) whose argmax is (h,w) of dy[b,ho,wo,c]  (gather, no atomics)
template <int NW>
__global__ __launch_bounds__(256) void maxpool3x3_bwd_kernel(const uint16_t* __restrict__ dy, int dy_cs,
                                                             const uint8_t* __restrict__ arg, uint16_t* __restrict__ dx,
                                                             int dx_cs, int B, int H, int W, int C, int Ho, int Wo, int stride,
                                                             int accumulate) {
  const int c8n = C >> 3;
  const long total = (long)B * H * W * c8n;
  for (long gi = (long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long)gridDim.x * 256) {
    const int c0 = (int)(gi % c8n) * 8;
    long t = gi / c8n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    // Position (h, w) lies in at most NW x NW windows (NW = 2 for stride >= 2, 3 for stride 1): their gradients, argmax bytes
    // and the old dx are all requested before the first is used (clamped address + select instead of data-dependent loops:
    // those compile to one branch, load and s_waitcnt per window).
    const int ho_lo = h >= 2 ? (h - 2 + stride - 1) / stride : 0, ho_hi = min(h / stride, Ho - 1);
    const int wo_lo = w >= 2 ? (w - 2 + stride - 1) / stride : 0, wo_hi = min(w / stride, Wo - 1);
    uint16_t* dst = dx + (((long)b * H + h) * W + w) * dx_cs + c0;
    uint4 gv[NW * NW], oldv = make_uint4(0, 0, 0, 0);
    uint2 av[NW * NW];
    bool ok[NW * NW];
    unsigned want[NW * NW];
#pragma unroll
    for (int t = 0; t < NW * NW; ++t) {
      const int ho = ho_lo + t / NW, wo = wo_lo + t % NW;
      ok[t] = ho <= ho_hi && wo <= wo_hi;
      want[t] = (unsigned)((h - ho * stride) * 3 + (w - wo * stride));
      const long o = ok[t] ? ((long)b * Ho + ho) * Wo + wo : 0L;
      av[t] = *reinterpret_cast<const uint2*>(arg + o * C + c0);
      gv[t] = *reinterpret_cast<const uint4*>(dy + o * dy_cs + c0);
    }
    if (accumulate) oldv = *reinterpret_cast<const uint4*>(dst);
    float acc[8];
    unpack8(oldv, acc);
#pragma unroll
    for (int t = 0; t < NW * NW; ++t) {
      float g[8];
      unpack8(gv[t], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned aj = ((j < 4 ? av[t].x : av[t].y) >> (8 * (j & 3))) & 0xffu;
        acc[j] += (ok[t] && aj == want[t]) ? g[j] : 0.f;
      }
    }
    *reinterpret_cast<uint4*>(dst) = pack8(acc);
  }
}

DTF_API int dtf_maxpool3x3_bwd(const void* dy, int dy_cs, const void* argmax, void* dx, int dx_cs, int B, int H, int W,
                               int C, int stride, int accumulate, hipStream_t st) {
  if (B <= 0 || C % 8 || dy_cs % 8 || dx_cs % 8 || H < 3 || W < 3 || stride < 1) return ERR_SHAPE;
  const int Ho = (H - 3) / stride + 1, Wo = (W - 3) / stride + 1;
  if (stride >= 2)
    hipLaunchKernelGGL(maxpool3x3_bwd_kernel<2>, dim3(ew_blocks((long)B * H * W * (C / 8))), dim3(256), 0, st,
                       (const uint16_t*)dy, dy_cs, (const uint8_t*)argmax, (uint16_t*)dx, dx_cs, B, H, W, C, Ho, Wo, stride,
                       accumulate);
  else
    hipLaunchKernelGGL(maxpool3x3_bwd_kernel<3>, dim3(ew_blocks((long)B * H * W * (C / 8))), dim3(256), 0, st,
                       (const uint16_t*)dy, dy_cs, (const uint8_t*)argmax, (uint16_t*)dx, dx_cs, B, H, W, C, Ho, Wo, stride,
                       accumulate);
  DTF_CHECK_LAUNCH();
  return 0;
}

// avg 3x3/1 SAME with TF's divisor (valid taps only): dx[h,w] (+)= sum_{|ho-h|<=1,|wo-w|<=1} dy[ho,wo] / cnt(ho,wo)
__global__ __launch_bounds__(256) void avgpool3x3_bwd_kernel(const uint16_t* __restrict__ dy, int dy_cs,
                                                             uint16_t* __restrict__ dx, int dx_cs, int B, int H, int W, int C,
                                                             int accumulate) {
  const int c8n = C >> 3;
  const long total = (long)B * H * W * c8n;
  for (long gi = (long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long)gridDim.x * 256) {
    const int c0 = (int)(gi % c8n) * 8;
    long t = gi / c8n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    // the nine candidate windows and the old dx are requested before the first is used (see maxpool3x3_bwd_kernel)
    uint16_t* dst = dx + (((long)b * H + h) * W + w) * dx_cs + c0;
    uint4 gv[9], oldv = make_uint4(0, 0, 0, 0);
    float inv[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int ho = h - 1 + t9 / 3, wo = w - 1 + t9 % 3;
      const bool ok = (unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W;
      const int nr = min(ho + 1, H - 1) - max(ho - 1, 0) + 1, nc = min(wo + 1, W - 1) - max(wo - 1, 0) + 1;
      inv[t9] = ok ? 1.f / (float)(nr * nc) : 0.f;
      gv[t9] = *reinterpret_cast<const uint4*>(dy + (ok ? (((long)b * H + ho) * W + wo) * dy_cs + c0 : 0L));
    }
    if (accumulate) oldv = *reinterpret_cast<const uint4*>(dst);
    float acc[8];
    unpack8(oldv, acc);
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      float g[8];
      unpack8(gv[t9], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(g[j], inv[t9], acc[j]);
    }
    *reinterpret_cast<uint4*>(dst) = pack8(acc);
  }
}

DTF_API int dtf_avgpool3x3_same_bwd(const void* dy, int dy_cs, void* dx, int dx_cs, int B, int H, int W, int C,
                                    int accumulate, hipStream_t st) {
  if (B <= 0 || C % 8 || dy_cs % 8 || dx_cs % 8 || H <= 0 || W <= 0) return ERR_SHAPE;
  hipLaunchKernelGGL(avgpool3x3_bwd_kernel, dim3(ew_blocks((long)B * H * W * (C / 8))), dim3(256), 0, st,
                     (const uint16_t*)dy, dy_cs, (uint16_t*)dx, dx_cs, B, H, W, C, accumulate);
  DTF_CHECK_LAUNCH();
  return 0;
}

// dx[b,p,c] = bf16(df[b,c] / HW)
__global__ __launch_bounds__(256) void global_avgpool_bwd_kernel(const float* __restrict__ df, uint16_t* __restrict__ dx,
                                                                 int B, int HW, int C) {
  const int c8n = C >> 3;
  const long total = (long)B * HW * c8n;
  const float inv = 1.f / (float)HW;
  for (long gi = (long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long)gridDim.x * 256) {
    const int c0 = (int)(gi % c8n) * 8;
    const long bp = gi / c8n;
    const int b = (int)(bp / HW);
    float f[8];
    ld8f(df + (long)b * C + c0, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= inv;
    *reinterpret_cast<uint4*>(dx + bp * C + c0) = pack8(f);
  }
}

DTF_API int dtf_global_avgpool_bwd(const float* df, void* dx, int B, int HW, int C, hipStream_t st) {
  if (B <= 0 || C % 8 || HW <= 0) return ERR_SHAPE;
  hipLaunchKernelGGL(global_avgpool_bwd_kernel, dim3(ew_blocks((long)B * HW * (C / 8))), dim3(256), 0, st, df,
                     (uint16_t*)dx, B, HW, C);
  DTF_CHECK_LAUNCH();
  return 0;
}
