// TF-1.x Adam over a flat fp32 range as a device function: the optimizer kernel (optim.hip) and the launch that fuses
// the update of the fc1 weights with the conv filter gradients (conv2_lds.hip) share it.
//   lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m,v EMA;  p -= lr_t*m/(sqrt(v)+eps)   (eps outside the bias correction)
// Workgroup `block` of `nblocks`, `nthreads` threads each, grid-strides over p.n/4 float4 with U of them in flight per lane.
// Returns the 1-based step t = *p.step + 1 it applied.
#pragma once
#include "common.h"
#include "dtf_structs.h"

// The last workgroup to finish advances the step counter (arrival ticket), so a captured graph replays unchanged.
// A skipped step (p.skip raised, see dtf_structs.h) leaves the counter where it is -- the same batch is retried -- clears
// that step's metric slot again, counts the skip and lowers the flag.
__device__ __forceinline__ void finish_step(const OptimParams& p, int t) {
  if (!p.ticket) return;  // a partial update of the arena: the launch covering the rest advances the step
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = atomicAdd(p.ticket, 1u);
    if (prev == gridDim.x - 1) {  // every workgroup has read *step (and the skip flag) and finished its updates
      *p.ticket = 0u;
      const bool skipped = p.skip && __hip_atomic_load(p.skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      const int next = skipped ? t - 1 : t;
      *p.step = next;
      if (skipped) {
        p.skip[1] += 1u;
        __hip_atomic_store(p.skip, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (p.metrics_ring && p.ring > 0) {
        p.metrics_ring[2 * (next % p.ring)] = 0.f;
        p.metrics_ring[2 * (next % p.ring) + 1] = 0.f;
      }
    }
  }
}

// b^t for integer t >= 0 by repeated squaring in double: <= 62 multiplications (relative error ~1e-14), a fraction of
// what the library pow costs -- the bias correction sits on the critical path of the short optimizer launch.
__device__ __forceinline__ void beta_powers(double b1, double b2, int t, double& b1p, double& b2p) {
  b1p = 1.0;
  b2p = 1.0;
  for (unsigned int e = (unsigned int)t; e; e >>= 1) {
    if (e & 1u) { b1p *= b1; b2p *= b2; }
    b1 *= b1;
    b2 *= b2;
  }
}

// MULTI: the gradient is the sum, in index order, of p.nsrc buffers p.Gsrcs[] (the ranks' exchange buffers of a one-shot
// all-reduce: the reduction happens in the optimizer's own loads, bit-identical on every rank).
// WSH: the parameters are double-buffered by step parity and replicated by this very update (OptimParams::Wsh).
template <bool MULTI = false, bool WSH = false>
__device__ __forceinline__ int adam_tf_body(const OptimParams& p, const int block, const int nblocks, const int nthreads) {
  const float b1 = p.beta1, b2 = p.beta2, ob1 = 1.f - p.beta1, ob2 = 1.f - p.beta2;
  const long n4 = p.n >> 2;
  const float* __restrict__ gsrc = p.Gsrc ? p.Gsrc : p.G;
  constexpr int U = 4;  // 16 independent 16-B loads in flight per lane before the first use
  const long stride = (long)nblocks * nthreads * U;
  long base = (long)block * nthreads * U + threadIdx.x;
  float4 g[U], m[U], v[U], w[U];
  const int par = WSH ? (*p.step & 1) : 0;
  const float* __restrict__ wsrc = WSH ? p.Wsh[par][p.wsh_rank] : p.P;
  auto load = [&](long at) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = at + (long)u * nthreads;
      const long j = i < n4 ? i : at;  // clamped address, result discarded below
      if (MULTI) {
        float4 part[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (r < p.nsrc) part[r] = reinterpret_cast<const float4*>(p.Gsrcs[r])[j];
        g[u] = part[0];
#pragma unroll
        for (int r = 1; r < 8; ++r)
          if (r < p.nsrc) { g[u].x += part[r].x; g[u].y += part[r].y; g[u].z += part[r].z; g[u].w += part[r].w; }
      } else {
        g[u] = reinterpret_cast<const float4*>(gsrc)[j];
      }
      m[u] = reinterpret_cast<const float4*>(p.M)[j];
      v[u] = reinterpret_cast<const float4*>(p.V)[j];
      w[u] = reinterpret_cast<const float4*>(wsrc)[j];
    }
  };
  // (agent-scope load: in the fused exchange + optimizer launch the flag is raised by another workgroup of this launch)
  const bool skip = p.skip && __hip_atomic_load(p.skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  if (skip) {  // do not apply: only discard the accumulated gradient segments
    const int t = *p.step + 1;
    for (; base < n4; base += stride)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = base + (long)u * nthreads;
        if (i < n4 && i * 4 >= p.zero_from) reinterpret_cast<float4*>(p.G)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    return t;
  }
  if (base < n4) load(base);
  // the step counter and the bias correction are needed for the update only: fetched and computed while the first
  // batch of loads is in flight
  const int t = *p.step + 1;
  double b1p, b2p;
  beta_powers((double)p.beta1, (double)p.beta2, t, b1p, b2p);
  const float lr_t = (float)((double)p.lr * sqrt(1.0 - b2p) / (1.0 - b1p));
  while (base < n4) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * nthreads;
      if (i >= n4) continue;
      float* gp = &g[u].x;
      float* mp = &m[u].x;
      float* vp = &v[u].x;
      float* wp = &w[u].x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float gg = gp[c] * p.grad_scale;
        mp[c] = b1 * mp[c] + ob1 * gg;
        vp[c] = b2 * vp[c] + ob2 * gg * gg;
        wp[c] -= lr_t * mp[c] / (sqrtf(vp[c]) + p.eps);
      }
      reinterpret_cast<float4*>(p.M)[i] = m[u];
      reinterpret_cast<float4*>(p.V)[i] = v[u];
      if (WSH) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (r < p.wsh_world) reinterpret_cast<float4*>(p.Wsh[par ^ 1][r])[i] = w[u];
      } else {
        reinterpret_cast<float4*>(p.P)[i] = w[u];
      }
      if (i * 4 >= p.zero_from) reinterpret_cast<float4*>(p.G)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    base += stride;
    if (base < n4) load(base);
  }
  return t;
}
