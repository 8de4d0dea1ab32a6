// Exact-fp32 MFMA GEMM core for gfx950 (reference precision: the reference trains in float32,
// /root/reference/demo1/train.py:28-41,103,120,132).
//
//   C[M,N] (+)= A[M,K] * B[K,N]      fp32 operands, fp32 accumulate, v_mfma_f32_16x16x4_f32
//
// The instruction is bit-for-bit a k-ordered fmaf chain (one rounding per product) and issues at 64 FLOP/clk/SIMD = 1/16 of
// the bf16 rate, so these kernels are MFMA-pipe bound, not LDS or HBM bound: the design goal is an even spread of MFMAs
// over the 1024 SIMDs, with the staging kept out of the way.  16x16 tiles (not 32x32): the batch of 100 rows pads to 112
// instead of 128.
//
// Operand fragments: lane l (r = l & 15, q = l >> 4) supplies A[row r][k = q] and B[k = q][col r] of a 4-deep k-step;
// C/D register i -> row 4q + i, col r (cdna_hip_programming.md §3).  WHICH four k of the slab a k-step covers is free as
// long as A and B agree.  Two LDS images:
//   F32RowK : source k-contiguous (src[r*ld + k]) -> LDS [R][BK+4]; a lane reads its 4 consecutive k of a 16-deep group with
//             ONE ds_read_b128, element j feeding MFMA j of the group: k = 16*g + 4*q + j
//   F32ColK : source row-contiguous (src[k*ld + r]) -> LDS [BK][R+4]; single-dword reads, 16 lanes contiguous.  In a kernel
//             where both operands are of this kind the natural order k = 4*s + q is used and BK only has to be a multiple of
//             4 (the filter-gradient GEMMs have K = batch = 100 = 25 k-steps: no padding to 112).
// Pitches: (BK+4)*4 B keeps the 16-byte alignment of the b128 reads; (R+4) dwords puts the two k-rows of a 32-lane read
// group 16 banks apart.
#pragma once
#include "gemm_core.h"

__device__ __forceinline__ f32x4 mfma16f32(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// src[(r0 + r)*ld + k]; needs ld % 4 == 0, K % 4 == 0, 16-byte aligned base
template <int R, int NT, int BK, bool IS_A>
struct F32RowK {
  static constexpr int PITCH = BK + 4, CPR = BK / 4, NCHUNK = R * CPR, PER = (NCHUNK + NT - 1) / NT;
  static constexpr int LDS = R * PITCH;
  static constexpr bool KMAJOR = false;
  f32x4 regs[PER];
  const float* src;
  long ld;
  int rows, r0, k0, kend;
  __device__ __forceinline__ void init(const GemmParams& p, int r0_, int) {
    src = reinterpret_cast<const float*>(IS_A ? p.A : p.B);
    ld = IS_A ? p.lda : p.ldb;
    rows = IS_A ? p.M : p.N;
    r0 = r0_;
  }
  __device__ __forceinline__ bool ok(int c) const {
    return (NCHUNK % NT == 0 || c < NCHUNK) && (r0 + c / CPR) < rows && (k0 + (c % CPR) * 4) < kend;
  }
  __device__ __forceinline__ void load(int k0_, int kend_, int tid) {
    k0 = k0_;
    kend = kend_;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      regs[i] = *reinterpret_cast<const f32x4*>(src + (ok(c) ? (long)(r0 + c / CPR) * ld + k0 + (c % CPR) * 4 : 0L));
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      if (NCHUNK % NT == 0 || c < NCHUNK) *reinterpret_cast<f32x4*>(lds + (c / CPR) * PITCH + (c % CPR) * 4) = ok(c) ? regs[i] : z;
    }
  }
  // element j <-> k = 16*g + 4*q + j
  static __device__ __forceinline__ f32x4 frag(const float* lds, int row, int q, int g) {
    return *reinterpret_cast<const f32x4*>(lds + row * PITCH + g * 16 + 4 * q);
  }
};

// src[k*ld + r0 + r]; needs ld % 4 == 0, rows % 4 == 0 (or the tail rows readable: see ok()), 16-byte aligned base
template <int R, int NT, int BK, bool IS_A>
struct F32ColK {
  static constexpr int PITCH = R + 4, CPR = R / 4, NCHUNK = BK * CPR, PER = (NCHUNK + NT - 1) / NT;
  static constexpr int LDS = BK * PITCH;
  static constexpr bool KMAJOR = true;
  f32x4 regs[PER];
  const float* src;
  long ld;
  int rows, r0, k0, kend;
  __device__ __forceinline__ void init(const GemmParams& p, int r0_, int) {
    src = reinterpret_cast<const float*>(IS_A ? p.A : p.B);
    ld = IS_A ? p.lda : p.ldb;
    rows = IS_A ? p.M : p.N;
    r0 = r0_;
  }
  // a 4-row chunk is taken whole or not at all: rows is a multiple of 4 for every caller (checked by the launchers)
  __device__ __forceinline__ bool ok(int c) const {
    return (NCHUNK % NT == 0 || c < NCHUNK) && (k0 + c / CPR) < kend && (r0 + (c % CPR) * 4) < rows;
  }
  __device__ __forceinline__ void load(int k0_, int kend_, int tid) {
    k0 = k0_;
    kend = kend_;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      regs[i] = *reinterpret_cast<const f32x4*>(src + (ok(c) ? (long)(k0 + c / CPR) * ld + r0 + (c % CPR) * 4 : 0L));
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      if (NCHUNK % NT == 0 || c < NCHUNK) *reinterpret_cast<f32x4*>(lds + (c / CPR) * PITCH + (c % CPR) * 4) = ok(c) ? regs[i] : z;
    }
  }
  static __device__ __forceinline__ f32x4 frag(const float* lds, int row, int q, int g) {
    const float* p = lds + (g * 16 + 4 * q) * PITCH + row;
    const f32x4 v = {p[0], p[PITCH], p[2 * PITCH], p[3 * PITCH]};
    return v;
  }
};

// ------------------------------------------------------------------ epilogues (16x16 C/D map)
#define DTF_EP32_FOREACH(body)                                                      \
  _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                               \
    const int row = mbase + tm * 16 + 4 * (lane >> 4) + i;                          \
    const int col = nbase + tn * 16 + (lane & 15);                                  \
    const float v = acc[tm][tn][i];                                                 \
    body                                                                            \
  }

struct Ep32Store {  // C[z][m][n] = acc (+bias) (relu)
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x4 (&acc)[TM][TN], int mbase, int nbase, int lane, int z) {
    float* C = reinterpret_cast<float*>(p.C) + (long)z * p.slab_stride;
    DTF_EP32_FOREACH(if (row < p.M && col < p.N) {
      float o = v + (p.bias ? p.bias[col] : 0.f);
      if (p.relu) o = fmaxf(o, 0.f);
      C[(long)row * p.ldc + col] = o;
    })
  }
};

struct Ep32Atomic {  // C[m][n] += acc
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x4 (&acc)[TM][TN], int mbase, int nbase, int lane, int) {
    float* C = reinterpret_cast<float*>(p.C);
    DTF_EP32_FOREACH(if (row < p.M && col < p.N) atomicAdd(&C[(long)row * p.ldc + col], v);)
  }
};

// ------------------------------------------------------------------ the kernel body
// As: AL::LDS floats, Bs: BL::LDS floats (16-byte aligned).  Workgroup = WM x WN waves, each (BM/WM) x (BN/WN).
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
__device__ __forceinline__ void dtf_gemm32_body(const GemmParams& p, const int m0, const int n0, const int z,
                                                float* __restrict__ As, float* __restrict__ Bs) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr bool PLAIN = AL::KMAJOR && BL::KMAJOR;
  static_assert(TM >= 1 && TN >= 1 && BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "tile");
  static_assert(PLAIN ? (BK % 4 == 0) : (BK % 16 == 0), "slab depth");
  static_assert(TM * TN >= 2, "the 16x16x4 f32 MFMA needs two independent accumulators to issue back to back");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, q = lane >> 4;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.f;

  AL al;
  BL bl;
  al.init(p, m0, tid);
  bl.init(p, n0, tid);
  al.load(kbeg, kend, tid);
  bl.load(kbeg, kend, tid);
  const int arow = wm * (BM / WM) + r, brow = wn * (BN / WN) + r;
  for (int k = kbeg; k < kend; k += BK) {
    __syncthreads();
    al.store(As, tid);
    bl.store(Bs, tid);
    __syncthreads();
    if (k + BK < kend) {
      al.load(k + BK, kend, tid);
      bl.load(k + BK, kend, tid);
    }
    if constexpr (PLAIN) {
#pragma unroll
      for (int s = 0; s < BK / 4; ++s) {
        float af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = As[(4 * s + q) * AL::PITCH + arow + a * 16];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[b] = Bs[(4 * s + q) * BL::PITCH + brow + b * 16];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = mfma16f32(af[a], bf[b], acc[a][b]);
      }
    } else {
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[a] = AL::frag(As, arow + a * 16, q, g);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[b] = BL::frag(Bs, brow + b * 16, q, g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = mfma16f32(af[a][j], bf[b][j], acc[a][b]);
      }
    }
  }
  EP::template run<TM, TN>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, z);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
__global__ __launch_bounds__(64 * WM * WN) void dtf_gemm32_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[AL::LDS + BL::LDS];
  dtf_gemm32_body<BM, BN, BK, WM, WN, AL, BL, EP>(p, blockIdx.x * BM, blockIdx.y * BN, blockIdx.z, smem, smem + AL::LDS);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
static inline int dtf_launch_gemm32(const GemmParams& p, int splits, hipStream_t st) {
  dim3 grid(dtf_cdiv(p.M, BM), dtf_cdiv(p.N, BN), splits);
  hipLaunchKernelGGL((dtf_gemm32_kernel<BM, BN, BK, WM, WN, AL, BL, EP>), grid, dim3(64 * WM * WN), 0, st, p);
  DTF_CHECK_LAUNCH();
  return 0;
}
