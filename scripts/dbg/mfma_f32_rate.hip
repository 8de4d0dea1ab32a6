// Micro-benchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 with 1 / 2 / 4 waves per SIMD and 10 independent
// accumulators per wave (the shape of the fp32 conv2 kernels' inner loops, without any memory traffic).
//   hipcc --offload-arch=gfx950 -O3 scripts/dbg/mfma_f32_rate.hip -o scripts/dbg/mfma_f32_rate && scripts/dbg/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void rate_kernel(float* out, int iters, unsigned long long* clk) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, sizeof(float) * 1024 * 1024);
  (void)hipMalloc(&clk, sizeof(unsigned long long) * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000, NACC = 10;
  for (int wps : {1, 2, 4}) {
    const int threads = 64 * 4 * wps;  // waves per SIMD x 4 SIMDs
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(rate_kernel<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; (void)hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double med = 0; for (int i = 0; i < 256; ++i) med += (double)h[i]; med /= 256;
    const double mfma_per_simd = (double)iters * NACC * wps;
    printf("%d wave(s)/SIMD: kernel %.1f us; wave 0 of a workgroup: %.0f shader-clock ticks for %d MFMAs; per SIMD %.0f MFMAs -> %.2f ns per MFMA (%.1f cycles at 2.4 GHz), %.1f TFLOP/s chip-wide\n",
           wps, ms * 1e3, med, iters * NACC, mfma_per_simd, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4,
           2048.0 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
