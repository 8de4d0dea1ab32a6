// C ABI launchers of the exact-fp32 dense GEMMs (f32_gemm.h): the fc layers of the MNIST CNN at the reference's precision
// (demo1/train.py:94-120: tf.matmul on float32).  Same roles and the same horizontal fusion as the bf16 launchers in
// gemm_api.hip; operands and results are fp32 everywhere.
#include <type_traits>
#include "f32_gemm.h"
#include "f32x3_gemm.h"
#include "dtf_api.h"
#include "shard_body.h"

namespace {
constexpr int ERR_SHAPE = -22;

inline GemmParams base_params32(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc,
                                int splits, int BK) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  const int ksteps = dtf_cdiv(K, BK);
  p.kchunk = dtf_cdiv(ksteps, splits) * BK;
  p.slab_stride = (long)M * ldc;
  return p;
}
inline int eff_splits32(const GemmParams& p) { return dtf_cdiv(p.K, p.kchunk); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// 112 x 64 tile: the batch of 100 pads to 7 MFMA rows of 16; each of the 4 waves owns 16 output columns x 7 row tiles
// forward: 8 waves, two sets of 4 sharing each slab (KW = 2, see dtf_gemm32_body)
using FwdA = F32RowK<112, 512, 32, true>;
using FwdB = F32ColK<64, 512, 32, false>;
using BdA = F32RowK<112, 256, 32, true>;
using BdB = F32RowK<64, 256, 32, false>;
// filter gradients: K = batch, both operands row-contiguous; 128 x 64 tile, 4 waves of 32 x 64
using DwA = F32ColK<128, 256, 20, true>;
using DwB = F32ColK<64, 256, 20, false>;
using DwBn = F32ColK<16, 256, 20, false>;  // N <= 16 (fc2: 10 classes, rows of the gradient padded to 16 floats)

// K in segments: p.A / p.B are tables of operand blocks of p.kseg k-rows each (the "gather" data-parallel schedule: one block
// per rank, see dtf_fc_bwd_trio_f32).  kseg is a multiple of BK, so a slab lies in one block and the choice of block is
// uniform over the workgroup.
template <int R, int NT, int BK, bool IS_A>
struct F32ColKSeg : F32ColK<R, NT, BK, IS_A> {
  using Base = F32ColK<R, NT, BK, IS_A>;
  const float* const* tab;
  const float* nxt;  // block pointer of the slab this register set loads next (fetched one load() early: a table read
  int seg, nxt_s;    // in front of a slab's loads would put its latency into every slab, +6.5 us on the launch)
  int segp;          // places per block in the k index space (GemmParams::ksegp)
  __device__ __forceinline__ void init(const GemmParams& p, int r0_, int tid) {
    Base::init(p, r0_, tid);
    tab = reinterpret_cast<const float* const*>(IS_A ? p.A : p.B);
    seg = p.kseg;
    segp = p.ksegp > 0 ? p.ksegp : p.kseg;
    nxt_s = -1;
    nxt = nullptr;
  }
  __device__ __forceinline__ void load(int k0_, int kend_, int tid) {
    const int s = k0_ / segp;
    this->src = s == nxt_s ? nxt : tab[s];
    Base::load(k0_ - s * segp, seg, tid);
    nxt_s = (k0_ + 2 * BK) / segp;  // (the two register sets of the body alternate slabs)
    if (k0_ + 2 * BK < kend_) nxt = tab[nxt_s];
    else nxt_s = -1;
  }
};
using DwAg = F32ColKSeg<128, 256, 20, true>;
using DwBg = F32ColKSeg<64, 256, 20, false>;
// bf16x6 forms of the filter-gradient roles (f32x3_gemm.h): k-steps of 32, the batch of 100 pads to 128 per operand block
using DwA3 = F32ColK<128, 256, 32, true>;
using DwB3 = F32ColK<64, 256, 32, false>;
using DwBn3 = F32ColK<16, 256, 32, false>;
using DwAg3 = F32ColKSeg<128, 256, 32, true>;
using DwBg3 = F32ColKSeg<64, 256, 32, false>;

// The bf16x6 fc1 forward is wave-specialised (f32x3_gemm.h: dtf_gemm32x3_ws_body): 4 multiplying waves + 4 FWD_SMUL staging
// waves.  Measured per slab by the stamps: staging 3 100 / 2 400 cycles with 4 / 8 staging waves against ~1 400 of multiplying
// (kernel 13.9 / 11.4 us).  The exact-fp32 form has 512 threads.
constexpr int FWD_SMUL = 3;
using FwdAs = F32RowK<112, 256 * FWD_SMUL, 32, true>;
using FwdBs = F32ColK<64, 256 * FWD_SMUL, 32, false>;

// one role of a launch: the exact-fp32 body or its bf16x6 form (BK = the loaders' slab depth)
template <bool X3, int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP, int KW = 1, int NSET = 2, int NSLAB = 0>
__device__ __forceinline__ void gemm_role(const GemmParams& p, const int m0, const int n0, const int z, float* smem) {
  if constexpr (X3)
    dtf_gemm32x3_body<BM, BN, BK, WM, WN, AL, BL, EP, KW, NSET, NSLAB>(p, m0, n0, z, reinterpret_cast<char*>(smem));
  else dtf_gemm32_body<BM, BN, BK, WM, WN, AL, BL, EP, 2, KW>(p, m0, n0, z, smem);
}
template <bool X3, int BM, int BN, int BK, class AL, class BL, int STAGES = 1>
constexpr int role_lds_floats() {  // (STAGES: LDS images of a slab, bf16x6 only)
  if constexpr (X3) return dtf_gemm32x3_lds<BM, BN, BK, AL, BL, STAGES>() / 4;
  else return dtf_gemm32_lds<BM, BN, AL, BL>();
}

// bf16x6 fc1 forward: register sets of staged slabs.  Measured: 7 sets (a workgroup's whole K chunk, 3136 / 14 splits = 7
// slabs of 32, requested before the first slab is multiplied) 13.0 -> 15.6 us: the launch is bound by the 35 MB its 224
// workgroups draw through L2 (X is re-read by all 16 column tiles), not by the latency of a slab.
constexpr int FWD_NSET = 2;

// the fc1 forward (112 x 64 tile, 8 waves) in both forms; SEL: W double-buffered by step parity; PUSH: a copy role in the z
// planes behind the K splits (dtf_fc_fwd_f32_push)
// (one workgroup per CU: the bf16x6 form keeps two LDS images of a slab, 2 x 45 KB of dynamic shared memory)
constexpr int FWD_STAGES = 2;
template <bool X3>
constexpr int fwd_lds_bytes() { return 4 * role_lds_floats<X3, 112, 64, 32, FwdA, FwdB, X3 ? FWD_STAGES : 1>(); }
template <bool X3>
__device__ __forceinline__ void fwd_role(const GemmParams& p, float* smem) {
  if constexpr (X3)
    // (7: fc1's 3136 / 14 splits = 7 slabs of 32 take the unrolled path)
    dtf_gemm32x3_ws_body<112, 64, 32, 1, 4, FwdAs, FwdBs, Ep32Store, 7, FWD_SMUL>(p, blockIdx.x * 112, blockIdx.y * 64, blockIdx.z,
                                                                       reinterpret_cast<char*>(smem));
  else
    dtf_gemm32_body<112, 64, 32, 1, 4, FwdA, FwdB, Ep32Store, 2, 2>(p, blockIdx.x * 112, blockIdx.y * 64, blockIdx.z, smem);
}
template <bool X3>
constexpr int fwd_threads() { return X3 ? 256 * (1 + FWD_SMUL) : 512; }
template <bool X3, bool SEL>
__global__ __launch_bounds__(fwd_threads<X3>()) void fc_fwd32_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  GemmParams q = p;
  if constexpr (SEL) {
    if (*p.bsel & 1) q.B = p.B_alt;  // (workgroup-uniform)
  }
  fwd_role<X3>(q, smem);
}
template <class K>
int allow_lds(K kernel, int bytes) {  // (dynamic shared memory above 64 KB has to be granted per kernel, once)
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
template <bool X3, bool SEL>
int launch_fc_fwd32(const GemmParams& p, int splits, hipStream_t st) {
  static int granted = allow_lds(fc_fwd32_kernel<X3, SEL>, fwd_lds_bytes<X3>());
  if (granted != 0) return granted;
  dim3 grid(dtf_cdiv(p.M, 112), dtf_cdiv(p.N, 64), splits);
  hipLaunchKernelGGL((fc_fwd32_kernel<X3, SEL>), grid, dim3(fwd_threads<X3>()), fwd_lds_bytes<X3>(), st, p);
  DTF_CHECK_LAUNCH();
  return 0;
}
}  // namespace

// out[z][M][N] (fp32 slabs) = X[M][K] (fp32 row-major) * W[K][N] (fp32, TF [in,out] layout)
// splits == 1: bias/relu (optional) are applied; splits > 1: raw partial sums per slab.
DTF_API int dtf_fc_fwd_f32(const float* X, const float* W, float* out, int M, int N, int K, int splits, const float* bias,
                           int relu, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 4 || N % 4 || splits < 1 || !aligned16(X) || !aligned16(W)) return ERR_SHAPE;
  GemmParams p = base_params32(X, W, out, M, N, K, K, N, N, splits, 32);
  const int s = eff_splits32(p);
  p.bias = (s == 1) ? bias : nullptr;
  p.relu = (s == 1) ? relu : 0;
  return dtf_f32_emulation() ? launch_fc_fwd32<true, false>(p, s, st) : launch_fc_fwd32<false, false>(p, s, st);
}

// The same with W double-buffered by step parity ("wshard" schedule, engine.hip: MnistEngine::wshard): W1 is read when *sel is odd.
DTF_API int dtf_fc_fwd_f32_sel(const float* X, const float* W0, const float* W1, const int* sel, float* out, int M, int N,
                               int K, int splits, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 4 || N % 4 || splits < 1 || !aligned16(X) || !aligned16(W0) || !W1 || !aligned16(W1) || !sel)
    return ERR_SHAPE;
  GemmParams p = base_params32(X, W0, out, M, N, K, K, N, N, splits, 32);
  p.bsel = sel;
  p.B_alt = W1;
  return dtf_f32_emulation() ? launch_fc_fwd32<true, true>(p, eff_splits32(p), st)
                             : launch_fc_fwd32<false, true>(p, eff_splits32(p), st);
}

// The fc1 forward with a copy role riding in spare workgroups ("ipc-gather2", ipc_allreduce.hip: dtf_ipc_push_desc): X (this
// rank's pool2, complete since the previous launch) is stored into every rank's landing block while the GEMM runs, so the
// operand exchange that follows the loss head only has dh left to move.  z planes [splits, splits + PUSH_PLANES) push.
namespace {
constexpr int PUSH_PLANES = 2;
template <bool X3>
__global__ __launch_bounds__(fwd_threads<X3>()) void fc_fwd32_push_kernel(const GemmParams p, const int splits, const ShardPush push) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.z >= splits) {  // (workgroup-uniform)
    const int per = gridDim.x * gridDim.y;
    shard_push_body(push, ((int)blockIdx.z - splits) * per + blockIdx.y * gridDim.x + blockIdx.x, PUSH_PLANES * per,
                    fwd_threads<X3>());
    return;
  }
  fwd_role<X3>(p, smem);
}
}  // namespace

DTF_API int dtf_fc_fwd_f32_push(const float* X, const float* W, float* out, int M, int N, int K, int splits,
                                const ShardPush* push, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 4 || N % 4 || splits < 1 || !aligned16(X) || !aligned16(W)) return ERR_SHAPE;
  if (!push || push->n4 <= 0 || !push->src || push->world < 1 || push->world > 8 || push->rank < 0 || push->rank >= push->world)
    return ERR_SHAPE;
  for (int r = 0; r < push->world; ++r)
    if (!push->dst[r] || !aligned16(push->dst[r])) return ERR_SHAPE;
  GemmParams p = base_params32(X, W, out, M, N, K, K, N, N, splits, 32);
  const int s = eff_splits32(p);
  dim3 grid(dtf_cdiv(p.M, 112), dtf_cdiv(p.N, 64), s + PUSH_PLANES);
  if (dtf_f32_emulation()) {
    static int granted = allow_lds(fc_fwd32_push_kernel<true>, fwd_lds_bytes<true>());
    if (granted != 0) return granted;
    hipLaunchKernelGGL((fc_fwd32_push_kernel<true>), grid, dim3(fwd_threads<true>()), fwd_lds_bytes<true>(), st, p, s, *push);
  } else {
    static int granted = allow_lds(fc_fwd32_push_kernel<false>, fwd_lds_bytes<false>());
    if (granted != 0) return granted;
    hipLaunchKernelGGL((fc_fwd32_push_kernel<false>), grid, dim3(512), fwd_lds_bytes<false>(), st, p, s, *push);
  }
  DTF_CHECK_LAUNCH();
  return 0;
}

#ifdef DTF_GEMM32_STAMPS
DTF_API int dtf_fc_fwd_f32_stamps(const float* X, const float* W, float* out, int M, int N, int K, int splits, void* stamps,
                                  hipStream_t st) {
  GemmParams p = base_params32(X, W, out, M, N, K, K, N, N, splits, 32);
  p.aux_out = stamps;
  if (dtf_f32_emulation()) return launch_fc_fwd32<true, false>(p, eff_splits32(p), st);
  return dtf_launch_gemm32<112, 64, 32, 1, 4, FwdA, FwdB, Ep32Store, 2, 2>(p, eff_splits32(p), st);
}
#endif

// dX[M][N] (fp32) (+)= dY[M][K] * W[N][K]^T, W fp32 [N][K].  splits > 1 accumulates with atomics into a pre-zeroed dX.
DTF_API int dtf_fc_bwd_data_f32(const float* dY, const float* W, float* dX, int M, int N, int K, int splits, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 4 || splits < 1 || !aligned16(dY) || !aligned16(W)) return ERR_SHAPE;
  GemmParams p = base_params32(dY, W, dX, M, N, K, K, K, N, splits, 32);
  const int s = eff_splits32(p);
  if (dtf_f32_emulation()) {
    if (s == 1) return dtf_launch_gemm32x3<112, 64, 32, 1, 4, BdA, BdB, Ep32Store>(p, 1, st);
    return dtf_launch_gemm32x3<112, 64, 32, 1, 4, BdA, BdB, Ep32Atomic>(p, s, st);
  }
  if (s == 1) return dtf_launch_gemm32<112, 64, 32, 1, 4, BdA, BdB, Ep32Store>(p, 1, st);
  return dtf_launch_gemm32<112, 64, 32, 1, 4, BdA, BdB, Ep32Atomic>(p, s, st);
}

// dW[M][N] (fp32) = X[K][M]^T * dY[K][N], X and dY fp32 row-major [batch][features] with row pitches ldx / ldy (multiples of
// 4 floats; a pitch wider than the feature count must be readable up to the next multiple of 4).
// splits > 1 accumulates atomically into a pre-zeroed dW.
DTF_API int dtf_fc_dw_f32(const float* X, int ldx, const float* dY, int ldy, float* dW, int M, int N, int K, int splits,
                          hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || ldx % 4 || ldy % 4 || ldx < M || ldy < N || splits < 1 || !aligned16(X) || !aligned16(dY))
    return ERR_SHAPE;
  if (dtf_f32_emulation()) {
    GemmParams p = base_params32(X, dY, dW, M, N, K, ldx, ldy, N, splits, 32);
    const int s = eff_splits32(p);
    if (N <= 16) {
      if (s == 1) return dtf_launch_gemm32x3<128, 16, 32, 4, 1, DwA3, DwBn3, Ep32Store>(p, 1, st);
      return dtf_launch_gemm32x3<128, 16, 32, 4, 1, DwA3, DwBn3, Ep32Atomic>(p, s, st);
    }
    if (s == 1) return dtf_launch_gemm32x3<128, 64, 32, 4, 1, DwA3, DwB3, Ep32Store>(p, 1, st);
    return dtf_launch_gemm32x3<128, 64, 32, 4, 1, DwA3, DwB3, Ep32Atomic>(p, s, st);
  }
  GemmParams p = base_params32(X, dY, dW, M, N, K, ldx, ldy, N, splits, 20);
  const int s = eff_splits32(p);
  if (N <= 16) {
    if (s == 1) return dtf_launch_gemm32<128, 16, 20, 4, 1, DwA, DwBn, Ep32Store>(p, 1, st);
    return dtf_launch_gemm32<128, 16, 20, 4, 1, DwA, DwBn, Ep32Atomic>(p, s, st);
  }
  if (s == 1) return dtf_launch_gemm32<128, 64, 20, 4, 1, DwA, DwB, Ep32Store>(p, 1, st);
  return dtf_launch_gemm32<128, 64, 20, 4, 1, DwA, DwB, Ep32Atomic>(p, s, st);
}

// The three GEMMs that follow the loss head in ONE launch (as dtf_fc_bwd_trio): the split-K fc1 data gradient and the
// fc1 / fc2 filter gradients.
namespace {
struct GemmTrio32 {
  GemmParams q, d0, d1;
  int nblk_q, q_mt, q_nt;  // data-gradient blocks come first: m-tile fastest, then n-tile, then K split
  int nblk0, mt0, mt1;     // then problem 0 (fc1 dW), then problem 1 (fc2 dW)
  // "gather" data parallelism (ipc_allreduce.hip): problem 0 sums over the operand pairs of all ranks instead of being
  // all-reduced afterwards; d0.K = granks * batch, d0.kseg = batch
  const float* gA[8];
  const float* gB[8];
  int ksets;  // shard modes: K sets of problem 0 (see the kernel)
  // optional bias gradients as column sums, in row order (stores, no atomics): db0[j] = sum_m dYq[m][j] (fc1), db1[j] = sum_k
  // dY1[k][j] (fc2); a few workgroups behind problem 1.  The loss head then adds nothing to them (its 100 workgroups' atomics
  // on the same 65 cache lines cost 2.4 us of its 10.3).
  float* db0;
  float* db1;
  int nblk1;
  // optional, with the above: columns 12 / 13 of dY1 (pitch 16) hold the rows' loss and hit count (head_kernel): their sums
  // are added to the step's metrics slot
  float* metrics;
  const int* step_ptr;
  int ring;
};
// Slab counts for the straight-line path of the bf16x6 body (f32x3_gemm.h; 0 = rolled loop).  Measured with the usual
// counts (6 for the fc1 data gradient, 4 for the filter gradients): straight-line code keeps both register sets' loads in
// flight over more of the program, 226 VGPRs instead of 168 -> two workgroups per CU instead of three, launch 19.2 -> 25.4 us
// (at the 168 cap it spills 280 bytes per lane).  Three resident workgroups hide a slab's round trip better than one
// workgroup's counted waits do.
constexpr int TRIO_Q_NSLAB = 0, TRIO_D_NSLAB = 0;
template <bool X3>
constexpr int trio32_lds() {
  return role_lds_floats<X3, 112, 64, 32, BdA, BdB>() > role_lds_floats<X3, 128, 64, X3 ? 32 : 20, DwA, DwB>()
             ? role_lds_floats<X3, 112, 64, 32, BdA, BdB>() : role_lds_floats<X3, 128, 64, X3 ? 32 : 20, DwA, DwB>();
}

// (the mode is a template parameter for the reason given at fc_bwd_trio_kernel in gemm_api.hip)
// MODE 0: this rank's operands; 1 ("gather"): K runs over all ranks' operand pairs; 2 / 3 ("shard"): problem 0 is this
// rank's row slice of the global gradient -- per-rank MFMA work independent of the number of ranks (shard_body.h).  3: one
// K set, plain stores; 2: several K sets of ranks_per_set source ranks each (enough workgroups to fill the chip at 8 ranks),
// accumulated with atomics into a pre-zeroed slice.
// X3: every role as bf16x6 (f32x3_gemm.h); the filter-gradient roles then walk K = batch in slabs of 32 (operand blocks
// padded to 128 places, GemmParams::ksegp).
template <int MODE, bool X3>
__global__ __launch_bounds__(256, 3) void fc_bwd_trio_f32_kernel(const GemmTrio32 g) {
  __shared__ __attribute__((aligned(16))) float smem[trio32_lds<X3>()];
  constexpr int DBK = X3 ? 32 : 20;
  using TA = std::conditional_t<X3, DwA3, DwA>;
  using TB = std::conditional_t<X3, DwB3, DwB>;
  using TBn = std::conditional_t<X3, DwBn3, DwBn>;
  using TAg = std::conditional_t<X3, DwAg3, DwAg>;
  using TBg = std::conditional_t<X3, DwBg3, DwBg>;
  const int id = blockIdx.x;
  if (id < g.nblk_q) {
    const int mt = id % g.q_mt, r = id / g.q_mt;
    if constexpr (MODE == 2 || MODE == 3) {
      if (g.q.bsel) {  // "wshard": W_fc1 double-buffered by step parity (workgroup-uniform choice)
        GemmParams q = g.q;
        if (*g.q.bsel & 1) q.B = g.q.B_alt;
        gemm_role<X3, 112, 64, 32, 1, 4, BdA, BdB, Ep32Atomic, 1, 2, TRIO_Q_NSLAB>(q, mt * 112, (r % g.q_nt) * 64, r / g.q_nt, smem);
        return;
      }
    }
    gemm_role<X3, 112, 64, 32, 1, 4, BdA, BdB, Ep32Atomic, 1, 2, TRIO_Q_NSLAB>(g.q, mt * 112, (r % g.q_nt) * 64, r / g.q_nt, smem);
  } else if (id < g.nblk_q + g.nblk0) {
    const int k = id - g.nblk_q;
    if constexpr (MODE == 2 || MODE == 3) {
      // K sets of d0.kchunk = ranks_per_set x batch k-rows: set z multiplies the operand blocks of its source ranks
      const int per = g.nblk0 / g.ksets, z = k / per, kk = k % per;
      GemmParams d = g.d0;
      d.A = g.gA;
      d.B = g.gB;
      if constexpr (MODE == 2)
        gemm_role<X3, 128, 64, DBK, 4, 1, TAg, TBg, Ep32Atomic>(d, (kk % g.mt0) * 128, (kk / g.mt0) * 64, z, smem);
      else
        gemm_role<X3, 128, 64, DBK, 4, 1, TAg, TBg, Ep32Store>(d, (kk % g.mt0) * 128, (kk / g.mt0) * 64, 0, smem);
    } else if constexpr (MODE == 1) {
      GemmParams d = g.d0;
      d.A = g.gA;
      d.B = g.gB;
      gemm_role<X3, 128, 64, DBK, 4, 1, TAg, TBg, Ep32Store>(d, (k % g.mt0) * 128, (k / g.mt0) * 64, 0, smem);
    } else
      gemm_role<X3, 128, 64, DBK, 4, 1, TA, TB, Ep32Store, 1, 2, TRIO_D_NSLAB>(g.d0, (k % g.mt0) * 128, (k / g.mt0) * 64, 0, smem);
  } else if (id < g.nblk_q + g.nblk0 + g.nblk1) {
    const int k = id - g.nblk_q - g.nblk0;
    gemm_role<X3, 128, 16, DBK, 4, 1, TA, TBn, Ep32Store, 1, 2, TRIO_D_NSLAB>(g.d1, (k % g.mt1) * 128, (k / g.mt1) * 16, 0, smem);
  } else {
    // bias gradients: one column per thread, rows summed in order (deterministic; every load is independent of the sum)
    const int w = id - g.nblk_q - g.nblk0 - g.nblk1, tid = threadIdx.x;
    const int nb0 = (g.q.K + 255) / 256;
    if (w < nb0) {
      const int j = w * 256 + tid;
      if (j < g.q.K) {
        const float* __restrict__ src = reinterpret_cast<const float*>(g.q.A) + j;
        float s = 0.f;
#pragma unroll 10
        for (int m = 0; m < g.q.M; ++m) s += src[(long)m * g.q.lda];
        g.db0[j] = s;
      }
    } else if (tid < g.d1.N || (g.metrics && (tid == 12 || tid == 13))) {
      const float* __restrict__ src = reinterpret_cast<const float*>(g.d1.B) + tid;
      float s = 0.f;
#pragma unroll 10
      for (int k = 0; k < g.d1.K; ++k) s += src[(long)k * g.d1.ldb];
      if (tid < g.d1.N) {
        g.db1[tid] = s;
      } else {  // (the only writer of the slot in a training step: the head left the per-row values to this launch)
        const int step = g.step_ptr ? *g.step_ptr : 0;
        g.metrics[(g.ring > 0 ? 2 * (step % g.ring) : 0) + tid - 12] += s;
      }
    }
  }
}

template <int MODE>
int launch_trio32(const GemmTrio32& g, int nblocks, hipStream_t st) {
  if (dtf_f32_emulation()) hipLaunchKernelGGL((fc_bwd_trio_f32_kernel<MODE, true>), dim3(nblocks), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((fc_bwd_trio_f32_kernel<MODE, false>), dim3(nblocks), dim3(256), 0, st, g);
  DTF_CHECK_LAUNCH();
  return 0;
}
}  // namespace

#ifdef DTF_GEMM32_STAMPS
static void* g_trio_stamps = nullptr;  // (host global: every role's workgroups stamp into [workgroup][64])
DTF_API void dtf_dbg_set_trio_stamps(void* p) { g_trio_stamps = p; }
#endif

// dW0[M0][N0] = X0^T dY0 (pitches M0 / N0), dW1[M1][N1] = X1^T dY1 (N1 <= 16, dY1 pitch ldy1) and
// dX[Mq][Nq] += dYq[Mq][Kq] * Wq[Nq][Kq]^T (split over K, dX pre-zeroed) in one launch.  K = batch.
// With granks > 0, dW0 = sum over r < granks of gA[r]^T gB[r] instead (operand pairs of the shapes of X0 / dY0, e.g. the
// ranks' exchanged activations; X0 / dY0 themselves are then unused); K must be a multiple of 20 then.
// db0 / db1 (both or neither): also the bias gradients db0[Kq] = column sums of dYq, db1[N1] = column sums of dY1 (stored).
DTF_API int dtf_fc_bwd_trio_f32_bias(const float* X0, const float* dY0, float* dW0, int M0, int N0, const float* X1,
                                     const float* dY1, int ldy1, float* dW1, int M1, int N1, int K, const float* dYq,
                                     const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits, const void* const* gA,
                                     const void* const* gB, int granks, float* db0, float* db1, float* metrics,
                                     int metrics_ring, const int* step_ptr, hipStream_t st) {
  if ((db0 != nullptr) != (db1 != nullptr) || (metrics && (!db0 || ldy1 < 16 || N1 > 12))) return ERR_SHAPE;
  if (M0 <= 0 || N0 <= 0 || M1 <= 0 || N1 <= 0 || K <= 0 || M0 % 4 || N0 % 4 || M1 % 4 || N1 > 16 || ldy1 % 4 || ldy1 < N1)
    return ERR_SHAPE;
  if (Mq <= 0 || Nq <= 0 || Kq <= 0 || Kq % 4 || splits < 2) return ERR_SHAPE;
  if (!aligned16(X0) || !aligned16(dY0) || !aligned16(X1) || !aligned16(dY1) || !aligned16(dYq) || !aligned16(Wq)) return ERR_SHAPE;
  if (granks < 0 || granks > 8 || (granks > 0 && (!gA || !gB || K % 20))) return ERR_SHAPE;
  GemmTrio32 g;
  g.ksets = 1;
  for (int r = 0; r < 8; ++r) {
    g.gA[r] = r < granks ? static_cast<const float*>(gA[r]) : nullptr;
    g.gB[r] = r < granks ? static_cast<const float*>(gB[r]) : nullptr;
    if (r < granks && (!g.gA[r] || !g.gB[r] || !aligned16(g.gA[r]) || !aligned16(g.gB[r]))) return ERR_SHAPE;
  }
  const bool x3 = dtf_f32_emulation() != 0;
  const int dbk = x3 ? 32 : 20, Kp = x3 ? dtf_cdiv(K, 32) * 32 : K;  // (bf16x6: operand blocks of Kp places, rows >= K zero)
  g.d0 = base_params32(X0, dY0, dW0, M0, N0, granks > 0 ? granks * Kp : K, M0, N0, N0, 1, dbk);
  g.d0.kseg = granks > 0 ? K : 0;
  g.d0.ksegp = granks > 0 ? Kp : 0;
  g.d1 = base_params32(X1, dY1, dW1, M1, N1, K, M1, ldy1, N1, 1, dbk);
  g.mt0 = dtf_cdiv(M0, 128);
  g.mt1 = dtf_cdiv(M1, 128);
  g.nblk0 = g.mt0 * dtf_cdiv(N0, 64);
  g.nblk1 = g.mt1 * dtf_cdiv(N1, 16);
  g.db0 = db0;
  g.db1 = db1;
  g.metrics = metrics;
  g.ring = metrics_ring;
  g.step_ptr = step_ptr;
  const int nblk1 = g.nblk1 + (db0 ? dtf_cdiv(Kq, 256) + 1 : 0);
  g.q = base_params32(dYq, Wq, dXq, Mq, Nq, Kq, Kq, Kq, Nq, splits, 32);
  const int s = eff_splits32(g.q);
  if (s < 2) return ERR_SHAPE;  // the atomic epilogue is the only one instantiated here
  g.q_mt = dtf_cdiv(Mq, 112);
  g.q_nt = dtf_cdiv(Nq, 64);
  g.nblk_q = g.q_mt * g.q_nt * s;
#ifdef DTF_GEMM32_STAMPS
  g.q.aux_out = g.d0.aux_out = g.d1.aux_out = g_trio_stamps;
#endif
  return granks > 0 ? launch_trio32<1>(g, g.nblk_q + g.nblk0 + nblk1, st) : launch_trio32<0>(g, g.nblk_q + g.nblk0 + nblk1, st);
}
DTF_API int dtf_fc_bwd_trio_f32(const float* X0, const float* dY0, float* dW0, int M0, int N0, const float* X1,
                                const float* dY1, int ldy1, float* dW1, int M1, int N1, int K, const float* dYq,
                                const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits, const void* const* gA,
                                const void* const* gB, int granks, hipStream_t st) {
  return dtf_fc_bwd_trio_f32_bias(X0, dY0, dW0, M0, N0, X1, dY1, ldy1, dW1, M1, N1, K, dYq, Wq, dXq, Mq, Nq, Kq, splits, gA, gB,
                                  granks, nullptr, nullptr, nullptr, 0, nullptr, st);
}

// The same launch for the "shard" data-parallel schedule (shard_body.h):
//   dW0_slice[M0_slice][N0] += sum over r < granks of gA[r]^T gB[r]
// gA[r]: rank r's [K][>= M0_slice] operand block with row pitch lda0, already offset to the slice's first column;
// gB[r]: rank r's [K][N0] block.  K (a multiple of 20) is split into sets of ranks_per_set source ranks: with more than one
// set the sets accumulate with atomics and dW0_slice must be zero on entry (the push that follows re-zeroes it); one set
// stores.  Problems 1 and q as in dtf_fc_bwd_trio_f32; Wq_alt / wq_sel (both or neither): Wq_alt replaces Wq when *wq_sel is odd.
DTF_API int dtf_fc_bwd_trio_f32_shard_bias(const float* X1, const float* dY1, int ldy1, float* dW1, int M1, int N1, int K,
                                           const float* dYq, const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits,
                                           const void* const* gA, const void* const* gB, int granks, int ranks_per_set,
                                           int lda0, float* dW0_slice, int M0_slice, int N0, const float* Wq_alt,
                                           const int* wq_sel, float* db0, float* db1, float* metrics, int metrics_ring,
                                           const int* step_ptr, hipStream_t st) {
  if ((db0 != nullptr) != (db1 != nullptr) || (metrics && (!db0 || ldy1 < 16 || N1 > 12))) return ERR_SHAPE;
  if ((Wq_alt != nullptr) != (wq_sel != nullptr) || (Wq_alt && !aligned16(Wq_alt))) return ERR_SHAPE;
  if (M0_slice <= 0 || N0 <= 0 || M1 <= 0 || N1 <= 0 || K <= 0 || M0_slice % 4 || N0 % 4 || M1 % 4 || N1 > 16 || ldy1 % 4 ||
      ldy1 < N1 || lda0 % 4 || lda0 < M0_slice)
    return ERR_SHAPE;
  if (Mq <= 0 || Nq <= 0 || Kq <= 0 || Kq % 4 || splits < 2) return ERR_SHAPE;
  if (!aligned16(X1) || !aligned16(dY1) || !aligned16(dYq) || !aligned16(Wq) || !aligned16(dW0_slice)) return ERR_SHAPE;
  if (granks < 1 || granks > 8 || !gA || !gB || ranks_per_set < 1 || K % 20) return ERR_SHAPE;
  GemmTrio32 g;
  g.ksets = dtf_cdiv(granks, ranks_per_set);
  for (int r = 0; r < 8; ++r) {
    g.gA[r] = r < granks ? static_cast<const float*>(gA[r]) : nullptr;
    g.gB[r] = r < granks ? static_cast<const float*>(gB[r]) : nullptr;
    if (r < granks && (!g.gA[r] || !g.gB[r] || !aligned16(g.gA[r]) || !aligned16(g.gB[r]))) return ERR_SHAPE;
  }
  const bool x3 = dtf_f32_emulation() != 0;
  const int dbk = x3 ? 32 : 20, Kp = x3 ? dtf_cdiv(K, 32) * 32 : K;  // (bf16x6: operand blocks of Kp places, rows >= K zero)
  g.d0 = base_params32(nullptr, nullptr, dW0_slice, M0_slice, N0, granks * Kp, lda0, N0, N0, 1, dbk);
  g.d0.kseg = K;
  g.d0.ksegp = Kp;
  g.d0.kchunk = ranks_per_set * Kp;  // (a multiple of the slab depth)
  g.d1 = base_params32(X1, dY1, dW1, M1, N1, K, M1, ldy1, N1, 1, dbk);
  g.mt0 = dtf_cdiv(M0_slice, 128);
  g.mt1 = dtf_cdiv(M1, 128);
  g.nblk0 = g.mt0 * dtf_cdiv(N0, 64) * g.ksets;
  g.nblk1 = g.mt1 * dtf_cdiv(N1, 16);
  g.db0 = db0;
  g.db1 = db1;
  g.metrics = metrics;
  g.ring = metrics_ring;
  g.step_ptr = step_ptr;
  const int nblk1 = g.nblk1 + (db0 ? dtf_cdiv(Kq, 256) + 1 : 0);
  g.q = base_params32(dYq, Wq, dXq, Mq, Nq, Kq, Kq, Kq, Nq, splits, 32);
  g.q.bsel = wq_sel;  // (optional: Wq double-buffered by the parity of *wq_sel)
  g.q.B_alt = Wq_alt;
  const int s = eff_splits32(g.q);
  if (s < 2) return ERR_SHAPE;
  g.q_mt = dtf_cdiv(Mq, 112);
  g.q_nt = dtf_cdiv(Nq, 64);
  g.nblk_q = g.q_mt * g.q_nt * s;
#ifdef DTF_GEMM32_STAMPS
  g.q.aux_out = g.d0.aux_out = g.d1.aux_out = g_trio_stamps;
#endif
  return g.ksets > 1 ? launch_trio32<2>(g, g.nblk_q + g.nblk0 + nblk1, st) : launch_trio32<3>(g, g.nblk_q + g.nblk0 + nblk1, st);
}
DTF_API int dtf_fc_bwd_trio_f32_shard(const float* X1, const float* dY1, int ldy1, float* dW1, int M1, int N1, int K,
                                      const float* dYq, const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits,
                                      const void* const* gA, const void* const* gB, int granks, int ranks_per_set, int lda0,
                                      float* dW0_slice, int M0_slice, int N0, const float* Wq_alt, const int* wq_sel,
                                      hipStream_t st) {
  return dtf_fc_bwd_trio_f32_shard_bias(X1, dY1, ldy1, dW1, M1, N1, K, dYq, Wq, dXq, Mq, Nq, Kq, splits, gA, gB, granks,
                                        ranks_per_set, lda0, dW0_slice, M0_slice, N0, Wq_alt, wq_sel, nullptr, nullptr, nullptr, 0,
                                        nullptr, st);
}
