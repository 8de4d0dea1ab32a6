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
template <int BM, int BN, int BK, class AL, class BL>
constexpr int dtf_gemm32x3_lds() {
  return X3Op<AL, BK>::BYTES + X3Op<BL, BK>::BYTES > BM * (BN + 4) * 4 ? X3Op<AL, BK>::BYTES + X3Op<BL, BK>::BYTES
                                                                       : BM * (BN + 4) * 4;
}

// NSET register sets of staged slabs: slabs i + 1 .. i + NSET - 1 are in flight while slab i is multiplied (a launch whose
// workgroups walk few slabs, like the fc1 forward with 7, takes them all at once and pays the memory latency once).
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int KW = 1, int NSET = 2>
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
    if (s == 0 || kbeg + s * BK < kend) {
      al[s].load(kbeg + s * BK, kend, tid);
      bl[s].load(kbeg + s * BK, kend, tid);
    }
  }
  for (int k = kbeg; k < kend; k += NSET * BK) {
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
      if (s > 0 && k + s * BK >= kend) break;  // (workgroup-uniform)
      if (k + s * BK > kbeg) __syncthreads();  // the previous slab has been multiplied
      AX::store(al[s], As, tid, NT);
      BX::store(bl[s], Bs, tid, NT);
      if (k + (s + NSET) * BK < kend) {
        al[s].load(k + (s + NSET) * BK, kend, tid);
        bl[s].load(k + (s + NSET) * BK, kend, tid);
      }
      __syncthreads();
#pragma unroll
      for (int g = 0; g < BK / 32; ++g) {
        // One A plane and two B planes live at a time (all nine would not fit next to the staged slabs' registers in the
        // 3-workgroups-per-CU launches): (a2 b0) (a0 b2) (a0 b0) | (a0 b1) (a1 b1) (a1 b0); two wave sets take a half each.
        bf16x8 a[TM], b0[TN], bx[TN];
        auto lda = [&](const int pl) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i] = AX::frag(As + pl * AX::PLANE, wm * (BM / WM) + i * 16, r, q, g);
        };
        auto ldb = [&](bf16x8 (&b)[TN], const int pl) {
#pragma unroll
          for (int j = 0; j < TN; ++j) b[j] = BX::frag(Bs + pl * BX::PLANE, wn * (BN / WN) + j * 16, r, q, g);
        };
#define X3G_PROD(BV)                                                              \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                  \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = mfma16bf(a[i], BV[j], acc[i][j]);
        if (KW == 1 || ks == 0) {
          lda(2); ldb(b0, 0); X3G_PROD(b0)
          lda(0); ldb(bx, 2); X3G_PROD(bx) X3G_PROD(b0)
        }
        if (KW == 1) {
          __builtin_amdgcn_sched_barrier(0);  // (keeps the second half's fragment reads from being hoisted: register room)
          ldb(bx, 1); X3G_PROD(bx)
          lda(1); X3G_PROD(bx) X3G_PROD(b0)
        } else if (ks == 1) {
          lda(0); ldb(bx, 1); X3G_PROD(bx)
          lda(1); ldb(b0, 0); X3G_PROD(bx) X3G_PROD(b0)
        }
#undef X3G_PROD
      }
    }
  }
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
