// Multi-tensor optimizers over the flat parameter arena (components C6/C7, kernel K14/K18).
// One launch updates every parameter: fp32 master weights, Adam slots and gradients live in four
// parallel flat arrays, so "8 x ApplyAdam" (demo1/train.py:132) is a single memory-bound stream.
//
// TF-1.x Adam:  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m,v EMA;  p -= lr_t*m/(sqrt(v)+eps)  (eps outside the
// bias correction).  The 1-based step t lives in device memory and is advanced by the last workgroup to
// finish (arrival ticket), so a captured hipGraph can be replayed without any host-side argument patch.
#include "common.h"
#include "dtf_structs.h"
#include "adam_body.h"


__global__ __launch_bounds__(256) void adam_tf_kernel(const OptimParams p) {
  const int t = adam_tf_body(p, blockIdx.x, gridDim.x, 256);
  finish_step(p, t);
}

__global__ __launch_bounds__(256) void sgd_kernel(const OptimParams p) {
  const int t = *p.step + 1;
  const float a = p.lr * p.grad_scale;
  const long n4 = p.n >> 2;
  const float* __restrict__ gsrc = p.Gsrc ? p.Gsrc : p.G;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 g = reinterpret_cast<const float4*>(gsrc)[i];
    float4 w = reinterpret_cast<const float4*>(p.P)[i];
    w.x -= a * g.x; w.y -= a * g.y; w.z -= a * g.z; w.w -= a * g.w;
    reinterpret_cast<float4*>(p.P)[i] = w;
    if (i * 4 >= p.zero_from) reinterpret_cast<float4*>(p.G)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  finish_step(p, t);
}

static int optim_grid(long n) {
  long blocks = (n / 4 + 1023) / 1024;  // 4 float4 per lane per pass
  if (blocks > 2048) blocks = 2048;  // memory-bound: cap and grid-stride (guide §6 G11)
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

DTF_API int dtf_adam_tf(const OptimParams* p, hipStream_t st) {
  if (p->n <= 0 || p->n % 4 || p->zero_from % 4 || !p->step) return -22;
  hipLaunchKernelGGL(adam_tf_kernel, dim3(optim_grid(p->n)), dim3(256), 0, st, *p);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_sgd(const OptimParams* p, hipStream_t st) {
  if (p->n <= 0 || p->n % 4 || p->zero_from % 4 || !p->step || !p->ticket) return -22;
  hipLaunchKernelGGL(sgd_kernel, dim3(optim_grid(p->n)), dim3(256), 0, st, *p);
  DTF_CHECK_LAUNCH();
  return 0;
}

// fp32 <-> bf16 casts of a gradient bucket (8 elements per lane; n % 8 == 0): the wire format of the fine-tune's optional
// bf16 gradient exchange.  The fp32 arena stays the accumulation and optimizer format.
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const float4 a = src[2 * i], b = src[2 * i + 1];
    dst[i] = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const uint4* __restrict__ src, float4* __restrict__ dst, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const uint4 v = src[i];
    dst[2 * i] = make_float4(bf2f(v.x & 0xffffu), bf2f(v.x >> 16), bf2f(v.y & 0xffffu), bf2f(v.y >> 16));
    dst[2 * i + 1] = make_float4(bf2f(v.z & 0xffffu), bf2f(v.z >> 16), bf2f(v.w & 0xffffu), bf2f(v.w >> 16));
  }
}

DTF_API int dtf_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t st) {
  if (n <= 0 || n % 8) return -22;
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((int)blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                     static_cast<uint4*>(dst), n / 8);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_cast_bf16_f32(const void* src, float* dst, long n, hipStream_t st) {
  if (n <= 0 || n % 8) return -22;
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((int)blocks), dim3(256), 0, st, static_cast<const uint4*>(src),
                     reinterpret_cast<float4*>(dst), n / 8);
  DTF_CHECK_LAUNCH();
  return 0;
}
