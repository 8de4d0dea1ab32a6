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

// ------------------------------------------------------------------ epilogues
// The accumulator tiles are first assembled as a [BM][BN+4] fp32 image in LDS (see the body) and then written out by the
// whole workgroup in full rows: 16 lanes cover 256 contiguous bytes.  (Straight from the 16x16 C/D fragments a store
// instruction covers four 64-byte pieces of four rows; measured on the fc1 forward: 8 000 cycles for 28 such stores per
// wave, a quarter of the kernel.)
struct Ep32Store {  // C[z][m][n] = acc (+bias) (relu)
  template <int BM, int BN>
  static __device__ __forceinline__ void tile(const GemmParams& p, const float* Cs, int m0, int n0, int z, int tid, int nt) {
    float* C = reinterpret_cast<float*>(p.C) + (long)z * p.slab_stride;
    const bool vec = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    for (int c = tid; c < BM * (BN / 4); c += nt) {
      const int row = c / (BN / 4), col = (c % (BN / 4)) * 4;
      if (m0 + row >= p.M || n0 + col >= p.N) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * (BN + 4) + col);
      if (p.bias) {  // (workgroup-uniform; the four loads are requested together: clamped address + select)
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = p.bias[n0 + col + j < p.N ? n0 + col + j : 0];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += n0 + col + j < p.N ? b[j] : 0.f;
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      float* dst = C + (long)(m0 + row) * p.ldc + n0 + col;
      if (vec && n0 + col + 3 < p.N) {
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n0 + col + j < p.N) dst[j] = v[j];
      }
    }
  }
};

struct Ep32Atomic {  // C[m][n] += acc
  template <int BM, int BN>
  static __device__ __forceinline__ void tile(const GemmParams& p, const float* Cs, int m0, int n0, int, int tid, int nt) {
    float* C = reinterpret_cast<float*>(p.C);
    for (int c = tid; c < BM * BN; c += nt) {  // one dword per lane: a wave instruction adds to 256 contiguous bytes
      const int row = c / BN, col = c % BN;
      if (m0 + row < p.M && n0 + col < p.N) atomicAdd(&C[(long)(m0 + row) * p.ldc + n0 + col], Cs[row * (BN + 4) + col]);
    }
  }
};

// LDS floats a workgroup of the body below needs
template <int BM, int BN, class AL, class BL>
constexpr int dtf_gemm32_lds() {
  return 2 * (AL::LDS + BL::LDS) > BM * (BN + 4) ? 2 * (AL::LDS + BL::LDS) : BM * (BN + 4);  // two slab images | output image
}

// ------------------------------------------------------------------ the kernel body
// smem: dtf_gemm32_lds() floats (16-byte aligned).  Workgroup = KW x WM x WN waves: WM x WN waves tile the BM x BN output,
// each (BM/WM) x (BN/WN), and KW such sets share every staged slab, set s multiplying its k-groups s, s+KW, ...  With one
// workgroup per CU (these GEMMs are a single wave of workgroups) KW = 2 puts two waves on every SIMD, so the LDS-read
// latencies, address arithmetic and staging of one run in the shadow of the other's MFMAs (one wave per SIMD: the
// multiply phase of a slab took 2 850 cycles for 1 792 cycles of MFMA, phase stamps of scripts/dbg/fc_fwd_stamps.py).
// The sets' partial sums are added in a fixed order in the LDS output image.
// Two register sets of staged slabs are in flight ahead of the one being multiplied (PF = 2).
// -DDTF_GEMM32_STAMPS (diagnostic builds only, scripts/dbg): thread 0 of every workgroup records the shader clock at the
// phase boundaries of its slab loop into p.aux_out[workgroup][64]
#ifdef DTF_GEMM32_STAMPS
#define DTF_STAMP(i) do { const int _i = (i); if (threadIdx.x == 0 && _i < 62) stamps[_i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DTF_STAMP(i) do { } while (0)
#endif
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int PF = 2, int KW = 1>
__device__ __forceinline__ void dtf_gemm32_body(const GemmParams& p, const int m0, const int n0, const int z,
                                                float* __restrict__ smem) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16, NT = 64 * WM * WN * KW;
  constexpr bool PLAIN = AL::KMAJOR && BL::KMAJOR;
  static_assert(TM >= 1 && TN >= 1 && BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "tile");
  static_assert(PLAIN ? (BK % 4 == 0) : (BK % 16 == 0), "slab depth");
  static_assert(TM * TN >= 2, "the 16x16x4 f32 MFMA needs two independent accumulators to issue back to back");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ks = wave / (WM * WN), wmn = wave % (WM * WN);
  const int wm = wmn / WN, wn = wmn % WN;
  const int r = lane & 15, q = lane >> 4;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
#ifdef DTF_GEMM32_STAMPS
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.aux_out) +
                               ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64;
  int si = 0;
  if (threadIdx.x == 0) {  // placement: HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID
    stamps[62] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    stamps[63] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
  }
#endif
  DTF_STAMP(si++);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.f;

  // Two LDS images of a slab: slab i + 1 is stored while slab i is being multiplied, so the loop has ONE barrier per slab and
  // the register -> LDS copy is off the critical path (single image: two barriers per slab and every wave idle around them
  // for ~60 % of the slab time, scripts/dbg/fc_trio_stamps.py).
  static_assert(PF == 2, "two register sets: slab i + 1 waits in registers while slab i + 2 / i + 3 is in flight");
  constexpr int STAGE = AL::LDS + BL::LDS;
  AL al[2];
  BL bl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    al[s].init(p, m0, tid);
    bl[s].init(p, n0, tid);
    if (s == 0 || kbeg + s * BK < kend) {
      al[s].load(kbeg + s * BK, kend, tid);
      bl[s].load(kbeg + s * BK, kend, tid);
    }
  }
  const int arow = wm * (BM / WM) + r, brow = wn * (BN / WN) + r;
  al[0].store(smem, tid);
  bl[0].store(smem + AL::LDS, tid);
  if (kbeg + 2 * BK < kend) {
    al[0].load(kbeg + 2 * BK, kend, tid);
    bl[0].load(kbeg + 2 * BK, kend, tid);
  }
  __syncthreads();
  DTF_STAMP(si++);
  for (int k = kbeg; k < kend; k += 2 * BK) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s > 0 && k + s * BK >= kend) break;  // (workgroup-uniform)
      const float* __restrict__ As = smem + s * STAGE;
      const float* __restrict__ Bs = As + AL::LDS;
      // slab (k + s BK) is in image s; the next one goes from register set 1 - s into image 1 - s (last read before the
      // barrier that ended the previous slab), and that register set is refilled with the slab after the next two
      if (k + (s + 1) * BK < kend) {
        al[1 - s].store(smem + (1 - s) * STAGE, tid);
        bl[1 - s].store(smem + (1 - s) * STAGE + AL::LDS, tid);
        if (k + (s + 3) * BK < kend) {
          al[1 - s].load(k + (s + 3) * BK, kend, tid);
          bl[1 - s].load(k + (s + 3) * BK, kend, tid);
        }
      }
      DTF_STAMP(si++);
      if constexpr (PLAIN) {
#pragma unroll
        for (int t0 = 0; t0 < BK / 4; t0 += KW) {
          const int t = t0 + ks;
          if (KW > 1 && t >= BK / 4) break;
          float af[TM], bf[TN];
#pragma unroll
          for (int a = 0; a < TM; ++a) af[a] = As[(4 * t + q) * AL::PITCH + arow + a * 16];
#pragma unroll
          for (int b = 0; b < TN; ++b) bf[b] = Bs[(4 * t + q) * BL::PITCH + brow + b * 16];
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = mfma16f32(af[a], bf[b], acc[a][b]);
        }
      } else {
#pragma unroll
        for (int g0 = 0; g0 < BK / 16; g0 += KW) {
          const int g = g0 + ks;
          if (KW > 1 && g >= BK / 16) break;
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
      __syncthreads();
      DTF_STAMP(si++);
    }
  }
  DTF_STAMP(si++);
  // ---- assemble the output image in LDS (the staged slabs are dead): set KW-1 stores, the others add in descending order
  float* Cs = smem;
#pragma unroll
  for (int s = KW - 1; s >= 0; --s) {
    __syncthreads();
    if (ks == s) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float* c = Cs + (wm * (BM / WM) + a * 16 + 4 * q + i) * (BN + 4) + wn * (BN / WN) + b * 16 + r;
            *c = (s == KW - 1) ? acc[a][b][i] : *c + acc[a][b][i];
          }
    }
  }
  __syncthreads();
  EP::template tile<BM, BN>(p, Cs, m0, n0, z, tid, NT);
  DTF_STAMP(si++);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int PF = 2, int KW = 1>
__global__ __launch_bounds__(64 * WM * WN * KW) void dtf_gemm32_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[dtf_gemm32_lds<BM, BN, AL, BL>()];
  dtf_gemm32_body<BM, BN, BK, WM, WN, AL, BL, EP, PF, KW>(p, blockIdx.x * BM, blockIdx.y * BN, blockIdx.z, smem);
}

// The same with a double-buffered B operand (GemmParams::bsel): the buffer is chosen by the parity of a device word, so a
// captured graph replays unchanged while the buffers alternate from step to step.
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int PF = 2, int KW = 1>
__global__ __launch_bounds__(64 * WM * WN * KW) void dtf_gemm32_bsel_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[dtf_gemm32_lds<BM, BN, AL, BL>()];
  GemmParams q = p;
  if (*p.bsel & 1) q.B = p.B_alt;  // (workgroup-uniform)
  dtf_gemm32_body<BM, BN, BK, WM, WN, AL, BL, EP, PF, KW>(q, blockIdx.x * BM, blockIdx.y * BN, blockIdx.z, smem);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int PF = 2, int KW = 1>
static inline int dtf_launch_gemm32(const GemmParams& p, int splits, hipStream_t st) {
  dim3 grid(dtf_cdiv(p.M, BM), dtf_cdiv(p.N, BN), splits);
  hipLaunchKernelGGL((dtf_gemm32_kernel<BM, BN, BK, WM, WN, AL, BL, EP, PF, KW>), grid, dim3(64 * WM * WN * KW), 0, st, p);
  DTF_CHECK_LAUNCH();
  return 0;
}
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int PF = 2, int KW = 1>
static inline int dtf_launch_gemm32_bsel(const GemmParams& p, int splits, hipStream_t st) {
  if (!p.bsel || !p.B_alt) return -22;
  dim3 grid(dtf_cdiv(p.M, BM), dtf_cdiv(p.N, BN), splits);
  hipLaunchKernelGGL((dtf_gemm32_bsel_kernel<BM, BN, BK, WM, WN, AL, BL, EP, PF, KW>), grid, dim3(64 * WM * WN * KW), 0, st, p);
  DTF_CHECK_LAUNCH();
  return 0;
}
