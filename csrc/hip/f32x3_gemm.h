// The fp32 GEMM core of f32_gemm.h on the bf16 matrix cores ("bf16x6", bf16x3.h):
//
//   C[M,N] (+)= A[M,K] * B[K,N]      fp32 operands in memory, three bf16 planes in LDS, six v_mfma_f32_16x16x32_bf16 per
//                                    32-deep k-step and tile, fp32 accumulate
//
// A drop-in for dtf_gemm32_body: the same loaders bring the fp32 slabs from memory into registers (F32RowK / F32ColK and
// everything derived from them: segmented K, double-buffered B), the same epilogues write the result.  What differs is the
// way through LDS: a staged chunk of four fp32 values is split once into three 8-byte plane pieces, and fragments are
//   RowK (source k-contiguous): plane [R][BK] bf16, pitch 2 BK + 16 B (an odd number of 16-byte units); one ds_read_b128
//     = a lane's 8 consecutive k
//   ColK (source row-contiguous): plane [BK][R] bf16 AS IT COMES, pitch 2 R + 64 B (four consecutive k-rows on distinct
//     banks); fragments with the transposing read ds_read_b64_tr_b16, two per 8 k
// One LDS image of a slab (two barriers per slab; the double image of the exact-fp32 body would need 2 x 45-49 KB), slabs
// i + 1 .. in flight in registers.  KW = 2: the two wave sets split the six plane products, three each.
#pragma once
#include "bf16x3.h"
#include "f32_gemm.h"

template <class L, int BK>
struct X3Op {
  static constexpr bool KMAJOR = L::KMAJOR;
  static constexpr int R = KMAJOR ? (L::PITCH - 4) : L::LDS / L::PITCH;  // rows / columns of the operand tile
  static constexpr int PITCH = KMAJOR ? 2 * R + 64 : 2 * BK + 16;        // bytes
  static constexpr int PLANE = (KMAJOR ? BK : R) * PITCH;
  static constexpr int BYTES = 3 * PLANE;
  static_assert(BK % 32 == 0, "k-steps of 32");

  static __device__ __forceinline__ void store(const L& l, char* lds, const int tid, const int nt) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < L::PER; ++i) {
      const int c = tid + nt * i;
      if (c < L::NCHUNK) {
        const Bf3Quad s = bf3_split4(l.ok(c) ? l.regs[i] : z);
        char* dst = lds + (c / L::CPR) * PITCH + (c % L::CPR) * 8;
        *reinterpret_cast<uint2*>(dst) = s.p0;
        *reinterpret_cast<uint2*>(dst + PLANE) = s.p1;
        *reinterpret_cast<uint2*>(dst + 2 * PLANE) = s.p2;
      }
    }
  }
  // fragment of the 16 rows / columns [base, base + 16), k-group g (32 deep), for lane (r, q)
  static __device__ __forceinline__ bf16x8 frag(const char* plane, const int base, const int r, const int q, const int g) {
    if constexpr (KMAJOR) {
      typedef short s16x4_t __attribute__((ext_vector_type(4)));
      typedef short s16x8_t __attribute__((ext_vector_type(8)));
      const char* p = plane + (32 * g + 8 * q + (r >> 2)) * PITCH + (base + 4 * (r & 3)) * 2;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 4 * PITCH));
      const s16x8_t v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v8);
    } else {
      return *reinterpret_cast<const bf16x8*>(plane + (base + r) * PITCH + (32 * g + 8 * q) * 2);
    }
  }
};

// LDS bytes a workgroup of the body below needs
template <int BM, int BN, int BK, class AL, class BL, int STAGES = 1>
constexpr int dtf_gemm32x3_lds() {
  return STAGES * (X3Op<AL, BK>::BYTES + X3Op<BL, BK>::BYTES) > BM * (BN + 4) * 4
             ? STAGES * (X3Op<AL, BK>::BYTES + X3Op<BL, BK>::BYTES) : BM * (BN + 4) * 4;
}

// NSET register sets of staged slabs: slabs i + 1 .. i + NSET - 1 are in flight while slab i is multiplied (a launch whose
// workgroups walk few slabs, like the fc1 forward with 7, takes them all at once and pays the memory latency once).
// NSLAB > 0: a K chunk of exactly NSLAB slabs (the launch's usual shape) runs as straight-line code, so that the waits for
// the staged registers are counted ("all but the N youngest loads"); in the rolled loop the compiler drains EVERY load in
// flight before each store (see dtf_gemm32x3_ws_body), which turns the two-slab lead of the loads into none.
// (A second LDS image was measured on the fc1 forward and changed nothing: every wave still splits, then multiplies.)
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int KW = 1, int NSET = 2, int NSLAB = 0>
__device__ __forceinline__ void dtf_gemm32x3_body(const GemmParams& p, const int m0, const int n0, const int z,
                                                  char* __restrict__ smem) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16, NT = 64 * WM * WN * KW;
  using AX = X3Op<AL, BK>;
  using BX = X3Op<BL, BK>;
  static_assert(TM >= 1 && TN >= 1 && BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "tile");
  static_assert(KW == 1 || KW == 2, "one wave set, or two that split the plane products");
  static_assert(AX::R == BM && BX::R == BN, "loader tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ks = wave / (WM * WN), wmn = wave % (WM * WN);
  const int wm = wmn / WN, wn = wmn % WN;
  const int r = lane & 15, q = lane >> 4;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

#ifdef DTF_GEMM32_STAMPS  // (diagnostic builds: see dtf_gemm32_body)
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.aux_out) +
                               ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64;
  int si = 0;
#endif
  DTF_STAMP(si++);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  char* As = smem;
  char* Bs = smem + AX::BYTES;
  AL al[NSET];
  BL bl[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s) {
    al[s].init(p, m0, tid);
    bl[s].init(p, n0, tid);
    // (unconditional: past kend the loaders read address 0 and zero-fill; a run-time condition would cost the counted waits)
    al[s].load(kbeg + s * BK, kend, tid);
    bl[s].load(kbeg + s * BK, kend, tid);
  }
  // one slab: register set S holds it; HASN: slab i + NSET exists (its loads go into the set that has just been stored)
#define X3G_PROD(BV)                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                               \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = mfma16bf(a[i_], BV[j_], acc[i_][j_]);
#define X3G_LDA(PL) _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) a[i_] = AX::frag(As + (PL) * AX::PLANE, wm * (BM / WM) + i_ * 16, r, q, g);
#define X3G_LDB(BV, PL) _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) BV[j_] = BX::frag(Bs + (PL) * BX::PLANE, wn * (BN / WN) + j_ * 16, r, q, g);
#define X3G_SLAB(S, FIRST, HASN, KK)                                                                             \
  do {                                                                                                           \
    if (!(FIRST)) __syncthreads(); /* the previous slab has been multiplied */                                   \
    DTF_STAMP(si++);                                                                                             \
    AX::store(al[S], As, tid, NT);                                                                               \
    BX::store(bl[S], Bs, tid, NT);                                                                               \
    if (HASN) {                                                                                                  \
      al[S].load((KK) + NSET * BK, kend, tid);                                                                   \
      bl[S].load((KK) + NSET * BK, kend, tid);                                                                   \
    }                                                                                                            \
    __syncthreads();                                                                                             \
    DTF_STAMP(si++);                                                                                             \
    _Pragma("unroll") for (int g = 0; g < BK / 32; ++g) {                                                        \
      /* One A plane and two B planes live at a time (all nine would not fit next to the staged slabs' registers in the */ \
      /* 3-workgroups-per-CU launches): (a2 b0) (a0 b2) (a0 b0) | (a0 b1) (a1 b1) (a1 b0); two wave sets take a half each. */ \
      bf16x8 a[TM], b0[TN], bx[TN];                                                                              \
      if (KW == 1 || ks == 0) {                                                                                  \
        X3G_LDA(2) X3G_LDB(b0, 0) X3G_PROD(b0)                                                                   \
        X3G_LDA(0) X3G_LDB(bx, 2) X3G_PROD(bx) X3G_PROD(b0)                                                      \
      }                                                                                                          \
      if (KW == 1) {                                                                                             \
        __builtin_amdgcn_sched_barrier(0); /* (keeps the second half's fragment reads from being hoisted: register room) */ \
        X3G_LDB(bx, 1) X3G_PROD(bx)                                                                              \
        X3G_LDA(1) X3G_PROD(bx) X3G_PROD(b0)                                                                     \
      } else if (ks == 1) {                                                                                      \
        X3G_LDA(0) X3G_LDB(bx, 1) X3G_PROD(bx)                                                                   \
        X3G_LDA(1) X3G_LDB(b0, 0) X3G_PROD(bx) X3G_PROD(b0)                                                      \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
  if (NSLAB > 0 && kend - kbeg > (NSLAB - 1) * BK && kend - kbeg <= NSLAB * BK) {  // (workgroup-uniform)
#pragma unroll
    for (int i = 0; i < (NSLAB > 0 ? NSLAB : 1); ++i) X3G_SLAB(i % NSET, i == 0, i + NSET < NSLAB, kbeg + i * BK);
  } else {
    for (int k = kbeg; k < kend; k += NSET * BK) {
#pragma unroll
      for (int s = 0; s < NSET; ++s) {
        if (s > 0 && k + s * BK >= kend) break;  // (workgroup-uniform)
        X3G_SLAB(s, k + s * BK == kbeg, k + (s + NSET) * BK < kend, k + s * BK);
      }
    }
  }
#undef X3G_SLAB
#undef X3G_LDA
#undef X3G_LDB
#undef X3G_PROD
  DTF_STAMP(si++);
  // ---- assemble the output image in LDS (the staged slab is dead): set KW-1 stores, the others add in descending order
  float* Cs = reinterpret_cast<float*>(smem);
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

// ------------------------------------------------------------------ wave-specialised form (one workgroup per CU)
// Workgroup = (1 + SMUL) x WM x WN waves: the first WM x WN MULTIPLY (every plane product of their tiles), the others STAGE (load,
// split, store into the other LDS image).  Why: phase stamps of the fc1 forward (scripts/dbg/fc_fwd_stamps.py) show a wave of
// the plain body spending ~1 250 cycles per slab splitting and storing and ~1 500 reading fragments and multiplying, one
// after the other, with or without a second LDS image -- every wave does both jobs and the barrier keeps all of them on the
// same one.  With the jobs on different waves of a SIMD the split's VALU work runs under the other wave's MFMAs.
// AL / BL are instantiated for the 64 x WM x WN x SMUL staging threads.  smem: 2 x (A + B images) bytes.
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int NSLAB = 0, int SMUL = 1>
__device__ __forceinline__ void dtf_gemm32x3_ws_body(const GemmParams& p, const int m0, const int n0, const int z,
                                                     char* __restrict__ smem) {
  // NC multiplying threads, NS = SMUL x NC staging threads (the split is the larger job: ~500 cycles per 4-value chunk and
  // wave by the stamps, 6 144 values per slab of the fc1 forward)
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16, NC = 64 * WM * WN, NS = SMUL * NC, NT = NC + NS;
  using AX = X3Op<AL, BK>;
  using BX = X3Op<BL, BK>;
  static_assert(TM >= 1 && TN >= 1 && BM % (WM * 16) == 0 && BN % (WN * 16) == 0, "tile");
  static_assert(AX::R == BM && BX::R == BN, "loader tile");
  constexpr int STAGE = AX::BYTES + BX::BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool stager = wave >= WM * WN;  // (wave-uniform)
  const int ptid = tid - NC;
  const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;
  const int r = lane & 15, q = lane >> 4;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
#ifdef DTF_GEMM32_STAMPS
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.aux_out) +
                               ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64;
  int si = 0, sj = 0;
#endif
  DTF_STAMP(si++);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  AL al[2];
  BL bl[2];
  if (stager) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      al[s].init(p, m0, ptid);
      bl[s].init(p, n0, ptid);
      // (unconditional: past kend the loaders read address 0 and zero-fill, and a run-time condition here would cost the
      // counted waits of everything that follows)
      al[s].load(kbeg + s * BK, kend, ptid);
      bl[s].load(kbeg + s * BK, kend, ptid);
    }
    AX::store(al[0], smem, ptid, NS);
    BX::store(bl[0], smem + AX::BYTES, ptid, NS);
    al[0].load(kbeg + 2 * BK, kend, ptid);
    bl[0].load(kbeg + 2 * BK, kend, ptid);
  }
  __syncthreads();
  DTF_STAMP(si++);
  // One slab step of either side; S = the LDS image being multiplied = the register set that is NOT stored now; HAS1 /
  // HAS3: slabs i + 1 / i + 3 exist.  Each side runs its own loop over the slabs (same number of barriers on both).
  // Why two loops and not one loop with an if: the waits for the staged registers are counted (s_waitcnt vmcnt(N): "all
  // but the N youngest loads have landed"), and the compiler can only count along straight-line code.  Where the two
  // sides' paths join after every slab, or a loop's back edge or a run-time "is there a next slab" sits in between, it
  // falls back to waiting for EVERY load in flight before each store, i.e. for the refill issued one slab ago: a full
  // memory round trip per slab (3 200 cycles by the stamps, for 1 400 cycles of multiplying).  Hence also the fully
  // unrolled path for the launch's usual slab count NSLAB.
#ifdef DTF_GEMM32_STAMPS
#define X3W_STAGER_STAMP() do { if (ptid == 0 && sj < 30) stamps[32 + sj++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define X3W_STAGER_STAMP() do { } while (0)
#endif
#define X3W_PROD(PA, PB)                                                          \
  _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                               \
      _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] = mfma16bf(a[i_][PA], b[j_][PB], acc[i_][j_]);
#define X3W_STAGE(S, HAS1, HAS3, KK)                                                                             \
  do {                                                                                                           \
    if (HAS1) {                                                                                                  \
      AX::store(al[1 - (S)], smem + (1 - (S)) * STAGE, ptid, NS);                                                \
      BX::store(bl[1 - (S)], smem + (1 - (S)) * STAGE + AX::BYTES, ptid, NS);                                    \
      if (HAS3) {                                                                                                \
        al[1 - (S)].load((KK) + 3 * BK, kend, ptid);                                                             \
        bl[1 - (S)].load((KK) + 3 * BK, kend, ptid);                                                             \
      }                                                                                                          \
    }                                                                                                            \
    X3W_STAGER_STAMP();                                                                                          \
    __syncthreads(); /* next image stored (and the current one multiplied, on the other side) */                 \
  } while (0)
#define X3W_MULTIPLY(S)                                                                                          \
  do {                                                                                                           \
    const char* As = smem + (S) * STAGE;                                                                         \
    const char* Bs = As + AX::BYTES;                                                                             \
    _Pragma("unroll") for (int g = 0; g < BK / 32; ++g) {                                                        \
      bf16x8 a[TM][3], b[TN][3];                                                                                 \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                         \
        const int pp = pl == 0 ? 2 : pl - 1; /* (plane 2 first: its product comes first) */                      \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                        \
          a[i_][pp] = AX::frag(As + pp * AX::PLANE, wm * (BM / WM) + i_ * 16, r, q, g);                          \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                                        \
          b[j_][pp] = BX::frag(Bs + pp * BX::PLANE, wn * (BN / WN) + j_ * 16, r, q, g);                          \
      }                                                                                                          \
      X3W_PROD(2, 0) X3W_PROD(0, 2) X3W_PROD(1, 1) X3W_PROD(1, 0) X3W_PROD(0, 1) X3W_PROD(0, 0)                  \
    }                                                                                                            \
    DTF_STAMP(si++); /* (products issued) */                                                                     \
    __syncthreads();                                                                                             \
    DTF_STAMP(si++);                                                                                             \
  } while (0)
  const bool usual = NSLAB > 0 && kend - kbeg == NSLAB * BK;  // (workgroup-uniform)
  if (stager) {
    if (usual) {
#pragma unroll
      for (int i = 0; i < (NSLAB > 0 ? NSLAB : 1); ++i) X3W_STAGE(i & 1, i + 1 < NSLAB, i + 3 < NSLAB, kbeg + i * BK);
    } else {
      for (int k = kbeg; k < kend; k += 2 * BK) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s > 0 && k + s * BK >= kend) break;
          X3W_STAGE(s, k + (s + 1) * BK < kend, k + (s + 3) * BK < kend, k + s * BK);
        }
      }
    }
  } else {
    if (usual) {
#pragma unroll
      for (int i = 0; i < (NSLAB > 0 ? NSLAB : 1); ++i) X3W_MULTIPLY(i & 1);
    } else {
      for (int k = kbeg; k < kend; k += 2 * BK) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s > 0 && k + s * BK >= kend) break;
          X3W_MULTIPLY(s);
        }
      }
    }
  }
#undef X3W_STAGE
#undef X3W_MULTIPLY
#undef X3W_PROD
#undef X3W_STAGER_STAMP
  // ---- the output image in LDS (the staged slabs are dead), written out by the whole workgroup
  float* Cs = reinterpret_cast<float*>(smem);
  if (!stager) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          Cs[(wm * (BM / WM) + a * 16 + 4 * q + i) * (BN + 4) + wn * (BN / WN) + b * 16 + r] = acc[a][b][i];
  }
  __syncthreads();
  EP::template tile<BM, BN>(p, Cs, m0, n0, z, tid, NT);
  DTF_STAMP(si++);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int KW = 1>
__global__ __launch_bounds__(64 * WM * WN * KW) void dtf_gemm32x3_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[dtf_gemm32x3_lds<BM, BN, BK, AL, BL>()];
  dtf_gemm32x3_body<BM, BN, BK, WM, WN, AL, BL, EP, KW>(p, blockIdx.x * BM, blockIdx.y * BN, blockIdx.z, smem);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int KW = 1>
static inline int dtf_launch_gemm32x3(const GemmParams& p, int splits, hipStream_t st) {
  dim3 grid(dtf_cdiv(p.M, BM), dtf_cdiv(p.N, BN), splits);
  hipLaunchKernelGGL((dtf_gemm32x3_kernel<BM, BN, BK, WM, WN, AL, BL, EP, KW>), grid, dim3(64 * WM * WN * KW), 0, st, p);
  DTF_CHECK_LAUNCH();
  return 0;
}
