// C ABI launchers for the MFMA GEMM / implicit-GEMM instantiations (see gemm_core.h).
// Every launcher validates the shapes/alignments its kernel assumes and returns a non-zero code
// instead of launching when they do not hold (a faulting kernel can reset the whole GPU host).
#include "gemm_core.h"

DTF_API int dtf_conv2_fwd_lds(const void* x, const float* W, const float* bias, void* out, void* amax, int B,
                              hipStream_t st);

namespace {
constexpr int ERR_SHAPE = -22;

inline GemmParams base_params(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb,
                              long ldc, int splits, int BK) {
  GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  const int ksteps = dtf_cdiv(K, BK);
  p.kchunk = dtf_cdiv(ksteps, splits) * BK;
  p.slab_stride = (long)M * ldc;
  return p;
}
inline int eff_splits(const GemmParams& p) { return dtf_cdiv(p.K, p.kchunk); }
}  // namespace

// out[z][M][N] (fp32 slabs) = X[M][K] (bf16|fp32, row-major) * W[K][N] (fp32, TF [in,out] layout)
// splits == 1: bias/relu (optional) are applied; splits > 1: raw partial sums per slab.
DTF_API int dtf_fc_fwd(const void* X, int x_is_f32, const float* W, float* out, int M, int N, int K, int splits,
                       const float* bias, int relu, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 8 || N % 2 || splits < 1) return ERR_SHAPE;
  GemmParams p = base_params(X, W, out, M, N, K, K, N, N, splits, 64);
  const int s = eff_splits(p);
  p.bias = (s == 1) ? bias : nullptr;
  p.relu = (s == 1) ? relu : 0;
  if (x_is_f32)
    return dtf_launch_gemm<128, 64, 64, 4, 1, LoadRowK<float, 128, 256, 64, true>, LoadColK<float, 64, 256, 64, false>, EpStoreF32>(p, s, st);
  return dtf_launch_gemm<128, 64, 64, 4, 1, LoadRowK<uint16_t, 128, 256, 64, true>, LoadColK<float, 64, 256, 64, false>, EpStoreF32>(p, s, st);
}

// dX[M][N] (fp32) (+)= dY[M][K] (bf16) * W[N][K]^T with W fp32 [N][K] (TF [in,out] layout seen from the input side).
// splits > 1 accumulates with fp32 atomics into a pre-zeroed dX.
DTF_API int dtf_fc_bwd_data(const void* dY, const float* W, float* dX, int M, int N, int K, int splits, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 8 || splits < 1) return ERR_SHAPE;
  GemmParams p = base_params(dY, W, dX, M, N, K, K, K, N, splits, 64);
  const int s = eff_splits(p);
  if (s == 1)
    return dtf_launch_gemm<128, 64, 64, 4, 1, LoadRowK<uint16_t, 128, 256, 64, true>, LoadRowK<float, 64, 256, 64, false>, EpStoreF32>(p, 1, st);
  return dtf_launch_gemm<128, 64, 64, 4, 1, LoadRowK<uint16_t, 128, 256, 64, true>, LoadRowK<float, 64, 256, 64, false>, EpAtomicF32>(p, s, st);
}

// dW[M][N] (fp32) = X[K][M]^T * dY[K][N]   (both bf16 row-major [batch][features]; K = batch)
// splits > 1 accumulates atomically into a pre-zeroed dW.
DTF_API int dtf_fc_dw(const void* X, int x_is_f32, const void* dY, float* dW, int M, int N, int K, int splits, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || M % 2 || N % 2 || splits < 1) return ERR_SHAPE;
  GemmParams p = base_params(X, dY, dW, M, N, K, M, N, N, splits, 128);
  const int s = eff_splits(p);
  if (x_is_f32) {
    if (s == 1)
      return dtf_launch_gemm<128, 64, 128, 4, 1, LoadColK<float, 128, 256, 128, true>, LoadColK<uint16_t, 64, 256, 128, false>, EpStoreF32>(p, 1, st);
    return dtf_launch_gemm<128, 64, 128, 4, 1, LoadColK<float, 128, 256, 128, true>, LoadColK<uint16_t, 64, 256, 128, false>, EpAtomicF32>(p, s, st);
  }
  if (s == 1)
    return dtf_launch_gemm<128, 64, 128, 4, 1, LoadColK<uint16_t, 128, 256, 128, true>, LoadColK<uint16_t, 64, 256, 128, false>, EpStoreF32>(p, 1, st);
  return dtf_launch_gemm<128, 64, 128, 4, 1, LoadColK<uint16_t, 128, 256, 128, true>, LoadColK<uint16_t, 64, 256, 128, false>, EpAtomicF32>(p, s, st);
}

// Two weight-gradient GEMMs (bf16 activations, K = batch, single K pass) in one launch:
//   dW0[M0][N0] = X0^T * dY0   and   dW1[M1][N1] = X1^T * dY1
DTF_API int dtf_fc_dw_pair(const void* X0, const void* dY0, float* dW0, int M0, int N0, const void* X1, const void* dY1,
                           float* dW1, int M1, int N1, int K, hipStream_t st) {
  if (M0 <= 0 || N0 <= 0 || M1 <= 0 || N1 <= 0 || K <= 0 || K > 128 || M0 % 2 || N0 % 2 || M1 % 2 || N1 % 2) return ERR_SHAPE;
  GemmParams p0 = base_params(X0, dY0, dW0, M0, N0, K, M0, N0, N0, 1, 128);
  GemmParams p1 = base_params(X1, dY1, dW1, M1, N1, K, M1, N1, N1, 1, 128);
  return dtf_launch_gemm_pair<128, 64, 128, 4, 1, LoadColK<uint16_t, 128, 256, 128, true>, LoadColK<uint16_t, 64, 256, 128, false>, EpStoreF32>(p0, p1, st);
}

// The three GEMMs that follow the loss head in ONE launch: the fc1/fc2 filter gradients (single K pass, K = batch) and
// the split-K fc1 data gradient.  They are independent, each alone leaves most of the chip idle, and a kernel
// boundary or a cross-queue dependency costs 5-10 us on this ~120 us step (rocprofv3 timeline).
namespace {
struct GemmTrio {
  GemmPair pair;
  GemmParams q;             // data gradient: dX[M][N] += dY[M][K] * W[N][K]^T, split over K
  int nblk_q, q_mt, q_nt;   // its blocks come first: m-tile fastest, then n-tile, then K split
  // "gather" data parallelism (ipc_allreduce.hip): problem 0 of the pair sums over the operand pairs of all ranks,
  // read in place from the peers' exchange buffers, instead of being all-reduced afterwards
  const void* gA[8];
  const void* gB[8];
  int granks;  // 0 = off
};
using TrioDwA = LoadColK<uint16_t, 128, 256, 128, true>;
using TrioDwB = LoadColK<uint16_t, 64, 256, 128, false>;
using TrioBdA = LoadRowK<uint16_t, 128, 256, 64, true>;
using TrioBdB = LoadRowK<float, 64, 256, 64, false>;

// (GATHER is a template parameter, not a run-time branch: with both bodies in one kernel the register allocation of
// the plain path grew enough to lose a resident workgroup per CU, +9 us on the step)
template <bool GATHER>
__global__ __launch_bounds__(256, 2) void fc_bwd_trio_kernel(const GemmTrio g) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[DTF_TN_LDS / 2];  // the largest role (>= (128 + 64) * (128 + 8))
  static_assert(DTF_TN_LDS / 2 >= (128 + 64) * (128 + 8), "LDS of the generic roles");
  const int id = blockIdx.x;
  if (id < g.nblk_q) {
    const int mt = id % g.q_mt, r = id / g.q_mt;
    dtf_gemm_body<128, 64, 64, 4, 1, TrioBdA, TrioBdB, EpAtomicF32>(g.q, mt * 128, (r % g.q_nt) * 64, r / g.q_nt, smem,
                                                                    smem + 128 * (64 + 8));
  } else {
    const int j = id - g.nblk_q;
    const int which = j >= g.pair.nblk0;
    const int k = which ? j - g.pair.nblk0 : j;
    const int mt = which ? g.pair.mt1 : g.pair.mt0;
    if (which == 0) {
      // fc1 filter gradient: tiles staged as they are, transposing LDS reads (dtf_tn_body_tr)
      const void* a1[1] = {g.pair.p[0].A};
      const void* b1[1] = {g.pair.p[0].B};
      dtf_tn_body_tr<EpStoreF32>(g.pair.p[0], GATHER ? g.gA : a1, GATHER ? g.gB : b1, GATHER ? g.granks : 1, (k % mt) * 128,
                                 (k / mt) * 64, reinterpret_cast<char*>(smem));
    } else
      dtf_gemm_body<128, 64, 128, 4, 1, TrioDwA, TrioDwB, EpStoreF32>(g.pair.p[which], (k % mt) * 128, (k / mt) * 64, 0, smem,
                                                                      smem + 128 * (128 + 8));
  }
}
}  // namespace

// dW0 = X0^T dY0, dW1 = X1^T dY1 (as dtf_fc_dw_pair) and dX[Mq][Nq] += dYq[Mq][Kq] * Wq[Nq][Kq]^T (as dtf_fc_bwd_data with
// splits > 1: dX pre-zeroed) in one launch.  With granks > 0, dW0 = sum over r < granks of gA[r]^T gB[r] instead (operand
// pairs of the same shapes as X0 / dY0, e.g. in peer GPUs' exchange buffers; X0 / dY0 themselves are then unused).
DTF_API int dtf_fc_bwd_trio(const void* X0, const void* dY0, float* dW0, int M0, int N0, const void* X1, const void* dY1,
                            float* dW1, int M1, int N1, int K, const void* dYq, const float* Wq, float* dXq, int Mq, int Nq,
                            int Kq, int splits, const void* const* gA, const void* const* gB, int granks, hipStream_t st) {
  if (M0 <= 0 || N0 <= 0 || M1 <= 0 || N1 <= 0 || K <= 0 || K > 128 || M0 % 8 || N0 % 8 || M1 % 2 || N1 % 2) return ERR_SHAPE;
  if (Mq <= 0 || Nq <= 0 || Kq <= 0 || Kq % 8 || splits < 2) return ERR_SHAPE;
  if (granks < 0 || granks > 8 || (granks > 0 && (!gA || !gB))) return ERR_SHAPE;
  GemmTrio g;
  g.granks = granks;
  for (int r = 0; r < 8; ++r) {
    g.gA[r] = r < granks ? gA[r] : nullptr;
    g.gB[r] = r < granks ? gB[r] : nullptr;
    if (r < granks && (!g.gA[r] || !g.gB[r])) return ERR_SHAPE;
  }
  g.pair.p[0] = base_params(X0, dY0, dW0, M0, N0, K, M0, N0, N0, 1, 128);
  g.pair.p[1] = base_params(X1, dY1, dW1, M1, N1, K, M1, N1, N1, 1, 128);
  g.pair.mt0 = dtf_cdiv(M0, 128);
  g.pair.mt1 = dtf_cdiv(M1, 128);
  g.pair.nblk0 = g.pair.mt0 * dtf_cdiv(N0, 64);
  const int npair = g.pair.nblk0 + g.pair.mt1 * dtf_cdiv(N1, 64);
  g.q = base_params(dYq, Wq, dXq, Mq, Nq, Kq, Kq, Kq, Nq, splits, 64);
  const int s = eff_splits(g.q);
  if (s < 2) return ERR_SHAPE;  // the atomic epilogue is the only one instantiated here
  g.q_mt = dtf_cdiv(Mq, 128);
  g.q_nt = dtf_cdiv(Nq, 64);
  g.nblk_q = g.q_mt * g.q_nt * s;
  if (granks > 0)
    hipLaunchKernelGGL(fc_bwd_trio_kernel<true>, dim3(g.nblk_q + npair), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL(fc_bwd_trio_kernel<false>, dim3(g.nblk_q + npair), dim3(256), 0, st, g);
  DTF_CHECK_LAUNCH();
  return 0;
}

static inline int fill_geom(GemmParams& p, int H, int W, int Cin, int cstride, int Ho, int Wo, int KH, int KW,
                            int stride, int pad_h, int pad_w) {
  p.g.H = H; p.g.W = W; p.g.Cin = Cin; p.g.cstride = cstride; p.g.Ho = Ho; p.g.Wo = Wo;
  p.g.KH = KH; p.g.KW = KW; p.g.stride = stride; p.g.pad_h = pad_h; p.g.pad_w = pad_w;
  if (Cin % 8 || cstride % 8 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return ERR_SHAPE;
  return 0;
}

// Fused conv (stride 1) + bias + ReLU + 2x2/2 max-pool.  x: bf16 NHWC [B,H,W,Cin]; Wt: fp32 HWIO
// [KH*KW*Cin][Cout]; out: bf16 [B,H/2,W/2,Cout]; argmax: uint8 same shape (may be null).
DTF_API int dtf_conv_bias_relu_pool(const void* x, const float* Wt, const float* bias, void* out, void* argmax,
                                    int B, int H, int W, int Cin, int Cout, int KH, int KW, hipStream_t st) {
  if (B <= 0 || H % 2 || W % 2 || Cout % 2) return ERR_SHAPE;
  if (H == 14 && W == 14 && Cin == 32 && Cout == 64 && KH == 5 && KW == 5)  // MNIST conv2: LDS-resident kernel
    return dtf_conv2_fwd_lds(x, Wt, bias, out, argmax, B, st);
  const int M = B * H * W, K = KH * KW * Cin;
  GemmParams p = base_params(x, Wt, out, M, Cout, K, 0, Cout, Cout, 1, 128);
  if (fill_geom(p, H, W, Cin, Cin, H, W, KH, KW, 1, (KH - 1) / 2, (KW - 1) / 2)) return ERR_SHAPE;
  p.bias = bias;
  p.aux_out = argmax;
  return dtf_launch_gemm<64, 64, 128, 2, 2, LoadConvA<64, 256, 128, 1>, LoadColK<float, 64, 256, 128, false>, EpBiasReluPool>(p, 1, st);
}

// Conv backward-data (stride 1, SAME) fused with the ReLU mask of the producing layer's pooled output:
//   dX[b,h,w,ci] = (mask[b,h,w,ci] > 0) * sum_{tap,co} dY[b,h+dy'-p,w+dx'-p,co] * W[flip(tap)][ci][co]
// dY: bf16 NHWC [B,H,W,Cout] (dense); Wt fp32 HWIO; mask: bf16 [B,H,W,Cin]; dX: fp32 [B,H,W,Cin].
DTF_API int dtf_conv_bwd_data_mask(const void* dY, const float* Wt, const void* mask, float* dX, int B, int H, int W,
                                   int Cin, int Cout, int KH, int KW, hipStream_t st) {
  if (B <= 0 || Cout % 8 || mask == nullptr) return ERR_SHAPE;
  const int M = B * H * W, K = KH * KW * Cout;
  GemmParams p = base_params(dY, Wt, dX, M, Cin, K, 0, 0, Cin, 1, 128);
  if (fill_geom(p, H, W, Cout, Cout, H, W, KH, KW, 1, (KH - 1) / 2, (KW - 1) / 2)) return ERR_SHAPE;
  p.aux_in = mask;
  return dtf_launch_gemm<64, 32, 128, 2, 1, LoadConvA<64, 128, 128, 0>, LoadConvWBwd<32, 128, 128>, EpMaskMulF32>(p, 1, st);
}

// Conv filter gradient: dW[tap][ci][co] (fp32 HWIO, pre-zeroed, atomics) += sum_pos X[pos+tap][ci] * dY[pos][co]
// x: bf16 NHWC [B,H,W,Cin]; dY: bf16 NHWC [B,H,W,Cout] (stride 1, SAME).
DTF_API int dtf_conv_dw(const void* x, const void* dY, float* dW, int B, int H, int W, int Cin, int Cout, int KH,
                        int KW, int splits, hipStream_t st) {
  if (B <= 0 || Cin % 2 || Cout % 2 || splits < 1) return ERR_SHAPE;
  const int M = KH * KW * Cin, K = B * H * W;
  GemmParams p = base_params(x, dY, dW, M, Cout, K, 0, Cout, Cout, splits, 64);
  if (fill_geom(p, H, W, Cin, Cin, H, W, KH, KW, 1, (KH - 1) / 2, (KW - 1) / 2)) return ERR_SHAPE;
  return dtf_launch_gemm<128, 64, 64, 4, 1, LoadConvDwA<128, 256, 64>, LoadColK<uint16_t, 64, 256, 64, false>, EpAtomicF32>(p, eff_splits(p), st);
}

// General inference conv for the Inception-v3 backbone: x bf16 NHWC (channel stride in_cstride, so the
// input may be a channel slice of a concat buffer), Wp: bf16 packed [Cout][KH*KW*Cin] (BN folded),
// bias fp32 [Cout]; out bf16 written at channel stride out_cstride (a slice of the concat buffer).
DTF_API int dtf_conv_bias_relu(const void* x, const void* Wp, const float* bias, void* out, int B, int H, int W,
                               int Cin, int in_cstride, int Cout, int out_cstride, int KH, int KW, int stride,
                               int pad_h, int pad_w, int Ho, int Wo, int relu, hipStream_t st) {
  if (B <= 0 || Cout <= 0) return ERR_SHAPE;
  const int M = B * Ho * Wo, K = KH * KW * Cin;
  GemmParams p = base_params(x, Wp, out, M, Cout, K, 0, K, out_cstride, 1, 64);
  if (fill_geom(p, H, W, Cin, in_cstride, Ho, Wo, KH, KW, stride, pad_h, pad_w)) return ERR_SHAPE;
  p.bias = bias;
  p.relu = relu;
  if (Cout <= 32)
    return dtf_launch_gemm<128, 32, 64, 4, 1, LoadConvA<128, 256, 64, 0>, LoadRowK<uint16_t, 32, 256, 64, false>, EpBiasReluBf16>(p, 1, st);
  if (M <= 256 * 64)
    return dtf_launch_gemm<64, 64, 64, 2, 2, LoadConvA<64, 256, 64, 0>, LoadRowK<uint16_t, 64, 256, 64, false>, EpBiasReluBf16>(p, 1, st);
  return dtf_launch_gemm<128, 64, 64, 4, 1, LoadConvA<128, 256, 64, 0>, LoadRowK<uint16_t, 64, 256, 64, false>, EpBiasReluBf16>(p, 1, st);
}
