// Retrain head (components C33, C36; kernels K16-K20): the new softmax layer trained on cached
// 2048-d Inception bottlenecks (retrain1/retrain.py:262-297).
//
// The whole step is two launches and never leaves the GPU:
//   fwd : gather B rows of the HBM-resident bottleneck cache by sampled index -> logits = x W + b ->
//         softmax ("final_result") -> (double) softmax cross-entropy -> accuracy -> dlogits
//   bwd : dW = x^T dlogits, db = colsum(dlogits) and, single-GPU, the SGD update fused in the same kernel
//         (multi-GPU: gradients are written out for the all-reduce and dtf_sgd applies them).
// The device-side sampler reproduces the reference's class-balanced sampling with replacement
// (random class, then random index in [0, 2^27) modulo the class size: retrain1/retrain.py:329-333).
#include "common.h"

namespace {
constexpr int ERR_SHAPE = -22;
constexpr int BN = 2048;  // bottleneck size (retrain1/retrain.py:31)
}  // namespace

struct RetrainHeadParams {
  const float* cache;     // [N][2048] fp32 bottlenecks
  const int* labels_all;  // [N] class id of every cache row
  const int* rows;        // [B] sampled cache rows
  const float* W;         // [2048][C]
  const float* b;         // [C]
  int B, C;
  float grad_scale;       // 1/B
  int double_softmax;
  int training;
  float* probs;           // [B][C]  (optional)
  float* dz;              // [B][C]  (training)
  int* pred;              // [B]     (optional)
  float* metrics;         // [2] loss sum, correct count (atomics)
  float* logits;          // [B][C]  (optional: the pre-activations, for the reference's per-step histograms)
};

__global__ __launch_bounds__(256) void retrain_head_fwd_kernel(const RetrainHeadParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [C] logits + [4*C] partials
  float* z = lds;
  float* part = lds + p.C;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long src = p.rows ? p.rows[row] : row;
  const float* x = p.cache + src * BN + tid * 8;
  const float4 x0 = reinterpret_cast<const float4*>(x)[0], x1 = reinterpret_cast<const float4*>(x)[1];
  const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
  for (int c = 0; c < p.C; ++c) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(xv[j], p.W[(long)(tid * 8 + j) * p.C + c], s);
    s = wave_sum(s);
    if (lane == 0) part[wave * p.C + c] = s;
  }
  __syncthreads();
  for (int c = tid; c < p.C; c += 256) {
    z[c] = part[c] + part[p.C + c] + part[2 * p.C + c] + part[3 * p.C + c] + p.b[c];
    if (p.logits) p.logits[(long)row * p.C + c] = z[c];
  }
  __syncthreads();
  if (wave != 0) return;
  // one wave does the softmax algebra; lanes stride over classes
  float m = -3.0e38f;
  for (int c = lane; c < p.C; c += 64) m = fmaxf(m, z[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < p.C; c += 64) s += __expf(z[c] - m);
  s = wave_sum(s);
  const float inv = 1.f / s;
  const int y = p.labels_all[src];
  // argmax (first max) over probabilities
  float best = -1.f;
  int arg = 0;
  for (int c = lane; c < p.C; c += 64) {
    const float pr = __expf(z[c] - m) * inv;
    if (pr > best) { best = pr; arg = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  float loss;
  if (p.double_softmax) {
    float qs = 0.f;
    for (int c = lane; c < p.C; c += 64) qs += __expf(__expf(z[c] - m) * inv - 1.f);
    qs = wave_sum(qs);
    const float qinv = 1.f / qs;
    float dot = 0.f;
    for (int c = lane; c < p.C; c += 64) {
      const float pr = __expf(z[c] - m) * inv;
      const float q = __expf(pr - 1.f) * qinv;
      dot += (q - (c == y ? 1.f : 0.f)) * p.grad_scale * pr;
    }
    dot = wave_sum(dot);
    const float py = __expf(z[y] - m) * inv;
    loss = -__logf(__expf(py - 1.f) * qinv);
    for (int c = lane; c < p.C; c += 64) {
      const float pr = __expf(z[c] - m) * inv;
      const float q = __expf(pr - 1.f) * qinv;
      const float g = (q - (c == y ? 1.f : 0.f)) * p.grad_scale;
      if (p.training) p.dz[(long)row * p.C + c] = pr * (g - dot);
      if (p.probs) p.probs[(long)row * p.C + c] = pr;
    }
  } else {
    loss = __logf(s) - (z[y] - m);  // log-sum-exp form: exact even when p[y] underflows
    for (int c = lane; c < p.C; c += 64) {
      const float pr = __expf(z[c] - m) * inv;
      if (p.training) p.dz[(long)row * p.C + c] = (pr - (c == y ? 1.f : 0.f)) * p.grad_scale;
      if (p.probs) p.probs[(long)row * p.C + c] = pr;
    }
  }
  // (loss, arg and y are wave-uniform) both totals in one atomic instruction: they share a cache line with every other row's
  if (lane < 2) atomicAdd(&p.metrics[lane], lane == 0 ? loss : (arg == y ? 1.f : 0.f));
  if (lane == 0 && p.pred) p.pred[row] = arg;
}

DTF_API int dtf_retrain_head_fwd(const RetrainHeadParams* p, hipStream_t st) {
  if (p->B <= 0 || p->C <= 0 || p->C > 2048 || (p->training && !p->dz)) return ERR_SHAPE;
  hipLaunchKernelGGL(retrain_head_fwd_kernel, dim3(p->B), dim3(256), sizeof(float) * 5 * p->C, st, *p);
  DTF_CHECK_LAUNCH();
  return 0;
}

// thread t -> (c = t / 2048, k = t % 2048): consecutive lanes read consecutive cache floats.
// fused_lr > 0: W/b are updated in place (SGD); otherwise gW/gb receive the gradients.
__global__ __launch_bounds__(256) void retrain_head_bwd_kernel(const float* __restrict__ cache, const int* __restrict__ rows,
                                                               const float* __restrict__ dz, int B, int C, float* W,
                                                               float* bias, float* gW, float* gb, float fused_lr) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t < (long)BN * C) {
    const int c = (int)(t / BN), k = (int)(t % BN);
    float g = 0.f;
    // eight batch rows requested together (row index -> cache row is a dependent pair of loads: one pair per iteration is
    // B dependent round trips, 31 us at batch 100); summed in batch order as before
    for (int b0 = 0; b0 < B; b0 += 8) {
      float xv[8], dv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u < B ? b0 + u : B - 1;
        const long src = rows ? rows[b] : b;
        xv[u] = cache[src * BN + k];
        dv[u] = dz[(long)b * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) g = b0 + u < B ? fmaf(xv[u], dv[u], g) : g;
    }
    if (fused_lr > 0.f) W[(long)k * C + c] -= fused_lr * g; else gW[(long)k * C + c] = g;
  } else if (t < (long)BN * C + C) {
    const int c = (int)(t - (long)BN * C);
    float g = 0.f;
    for (int b = 0; b < B; ++b) g += dz[(long)b * C + c];
    if (fused_lr > 0.f) bias[c] -= fused_lr * g; else gb[c] = g;
  }
}

DTF_API int dtf_retrain_head_bwd(const float* cache, const int* rows, const float* dz, int B, int C, float* W, float* bias,
                                 float* gW, float* gb, float fused_lr, hipStream_t st) {
  if (B <= 0 || C <= 0 || (fused_lr <= 0.f && (!gW || !gb))) return ERR_SHAPE;
  const long total = (long)BN * C + C;
  hipLaunchKernelGGL(retrain_head_bwd_kernel, dim3(dtf_cdiv(total, 256)), dim3(256), 0, st, cache, rows, dz, B, C, W, bias,
                     gW, gb, fused_lr);
  DTF_CHECK_LAUNCH();
  return 0;
}

// dx[b][k] = sum_c dz[b][c] * W[k][c]: the gradient w.r.t. the bottleneck, needed only when the backbone is fine-tuned
// (SURVEY §2.6 config 5).  thread -> (b, k); consecutive lanes write consecutive k.  `clear` (optional): two floats set to
// zero, the metric slot of the NEXT step (the fine-tune trainer alternates two slots, so no memset launch is needed).
__global__ __launch_bounds__(256) void retrain_head_dx_kernel(const float* __restrict__ dz, const float* __restrict__ W,
                                                              float* __restrict__ dx, int B, int C, float* __restrict__ clear) {
  if (clear && blockIdx.x == 0 && threadIdx.x < 2) clear[threadIdx.x] = 0.f;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)B * BN) return;
  const int b = (int)(t / BN), k = (int)(t % BN);
  float g = 0.f;
  for (int c = 0; c < C; ++c) g = fmaf(dz[(long)b * C + c], W[(long)k * C + c], g);
  dx[t] = g;
}

DTF_API int dtf_retrain_head_dx(const float* dz, const float* W, float* dx, int B, int C, float* clear, hipStream_t st) {
  if (B <= 0 || C <= 0) return ERR_SHAPE;
  hipLaunchKernelGGL(retrain_head_dx_kernel, dim3(dtf_cdiv((long)B * BN, 256)), dim3(256), 0, st, dz, W, dx, B, C, clear);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Class-balanced sampler with replacement (C33): rows[i] = offsets[c] + (r % counts[c]),
// c uniform over classes, r uniform in [0, 2^27).  Counter-based RNG keyed by (seed, draw counter).
__global__ __launch_bounds__(256) void sample_rows_kernel(const long* __restrict__ offsets, const long* __restrict__ counts,
                                                          int C, int n, unsigned long long seed, unsigned long long draw,
                                                          int* __restrict__ rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)draw, (uint32_t)(draw >> 32), 0x72747261u),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  // skip classes that have no image in this split (the reference would log a fatal error)
  int c = (int)(r.x % (uint32_t)C);
  for (int tries = 0; tries < C && counts[c] == 0; ++tries) c = (c + 1) % C;
  const long cnt = counts[c] > 0 ? counts[c] : 1;
  rows[i] = (int)(offsets[c] + (long)((r.y & 0x7FFFFFFu) % (unsigned long)cnt));
}

DTF_API int dtf_sizeof_retrain_head_params() { return (int)sizeof(RetrainHeadParams); }

// The same with the draw counter in device memory (a captured step graph replays unchanged): draw = *counter + 1, and the
// counter is advanced by a one-thread kernel behind the sampler (every block of the sampler has read it by then).
__global__ __launch_bounds__(256) void sample_rows_dev_kernel(const long* __restrict__ offsets, const long* __restrict__ counts,
                                                              int C, int n, unsigned long long seed,
                                                              const unsigned long long* __restrict__ counter,
                                                              int* __restrict__ rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long draw = *counter + 1ull;
  const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)draw, (uint32_t)(draw >> 32), 0x72747261u),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  int c = (int)(r.x % (uint32_t)C);
  for (int tries = 0; tries < C && counts[c] == 0; ++tries) c = (c + 1) % C;
  const long cnt = counts[c] > 0 ? counts[c] : 1;
  rows[i] = (int)(offsets[c] + (long)((r.y & 0x7FFFFFFu) % (unsigned long)cnt));
}
__global__ void bump_counter_kernel(unsigned long long* counter) { *counter += 1ull; }

DTF_API int dtf_sample_rows_dev(const long* offsets, const long* counts, int C, int n, unsigned long long seed,
                                unsigned long long* counter, int* rows, hipStream_t st) {
  if (C <= 0 || n <= 0 || !counter) return ERR_SHAPE;
  hipLaunchKernelGGL(sample_rows_dev_kernel, dim3(dtf_cdiv(n, 256)), dim3(256), 0, st, offsets, counts, C, n, seed, counter, rows);
  DTF_CHECK_LAUNCH();
  hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, st, counter);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_sample_rows(const long* offsets, const long* counts, int C, int n, unsigned long long seed,
                            unsigned long long draw, int* rows, hipStream_t st) {
  if (C <= 0 || n <= 0) return ERR_SHAPE;
  hipLaunchKernelGGL(sample_rows_kernel, dim3(dtf_cdiv(n, 256)), dim3(256), 0, st, offsets, counts, C, n, seed, draw, rows);
  DTF_CHECK_LAUNCH();
  return 0;
}
