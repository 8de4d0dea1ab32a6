// LDS-resident 5x5 convolution kernels for the MNIST conv2 layer (14x14x32 -> 14x14x64, SAME).
//
// The generic implicit-GEMM kernel (gemm_core.h) walks K in slabs, and at batch 100 every slab is a
// global-memory round trip on a one-wave-of-workgroups grid: the kernel time is a chain of ~7-13
// dependent latencies.  Here one workgroup owns one image (x one half of the output channels): the
// halo'd image (26 KB) and the filter slice (50 KB) are pulled into LDS with ONE burst of
// independent loads, after which the 50 k-steps are pure ds_read + MFMA.
//
//  * A fragments: ds_read_b128 straight from the pixel-major image (80-byte pixel stride: 64 B of
//    channels + 16 B pad so the 16-lane read groups spread over the banks); rows are enumerated in
//    2x2-pool-window order so the max-pool happens on four consecutive accumulator registers.
//  * B fragments: the fp32 HWIO filter is stored in LDS in its natural [k][n] order (converted to bf16
//    on the way in, coalesced float4 loads) and read with the gfx950 transposing LDS load
//    ds_read_b64_tr_b16 (two per fragment) — no transposed weight copy has to be maintained.
//    A 32-lane half reads 4 rows x 64 B = one full 256-B bank row: conflict-free without padding.
#include "common.h"
#include "conv1_dw_body.h"
#include "conv2_pack.h"
#include "adam_body.h"

namespace {
constexpr int ERR_SHAPE = -22;
constexpr int PIX = 80;                 // bytes per halo pixel (32 ch bf16 + pad)
constexpr int IMG_BYTES = 18 * 18 * PIX;
constexpr int WROW = 64;                // bytes per filter row in LDS (32 out-channels bf16)
constexpr int W_BYTES = 800 * WROW;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // (arrays of HIP's uint4 struct can land in scratch)

__device__ __forceinline__ bf16x8 tr_frag(const char* lds_row0) {
  // two transposing reads: k = k0..k0+3 and k0+4..k0+7 for this lane's column
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_row0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_row0 + 4 * WROW));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
}  // namespace

// grid (B, 2); 256 threads.  x bf16 [B,14,14,32]; Wf packed bf16 [2][800][32] (conv2_pack.h); out bf16 [B,49,64];
// amax uint8 [B,49,64]
__global__ __launch_bounds__(256) void conv2_fwd_lds_kernel(const uint16_t* __restrict__ x, const char* __restrict__ Wf,
                                                            const float* __restrict__ bias, uint16_t* __restrict__ out,
                                                            uint8_t* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img = smem;
  char* wl = smem + IMG_BYTES;
  const int b = blockIdx.x, n0 = blockIdx.y * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- one burst of loads: image chunks (6 per thread, halo -> zero) + filter slice (25 float4 per thread)
  uint4 iv[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = tid + 256 * i;  // 1296 payload chunks (324 pixels x 4)
    const int pix = c >> 2, q = c & 3;
    const int hy = pix / 18, hx = pix - hy * 18;
    const int y = hy - 2, xx = hx - 2;
    const bool ok = c < 1296 && (unsigned)y < 14u && (unsigned)xx < 14u;
    iv[i] = *reinterpret_cast<const uint4*>(x + (ok ? (((long)b * 14 + y) * 14 + xx) * 32 + q * 8 : 0L));
    if (!ok) iv[i] = make_uint4(0, 0, 0, 0);
  }
  u32x4 wv[13];  // this half's packed slice: 51 200 B = 3200 chunks of 16 B, already in LDS order
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(Wf + blockIdx.y * W_BYTES);
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    const int c = tid + 256 * j;
    wv[j] = wsrc[c < 3200 ? c : 0];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = tid + 256 * i;
    if (c < 1296) *reinterpret_cast<uint4*>(img + (c >> 2) * PIX + (c & 3) * 16) = iv[i];
  }
#pragma unroll
  for (int j = 0; j < 13; ++j) {
    const int c = tid + 256 * j;
    if (c < 3200) *reinterpret_cast<u32x4*>(wl + c * 16) = wv[j];
  }
  __syncthreads();

  // ---- per-lane fragment addresses
  const int r = lane & 31, h = lane >> 5;
  const int ntile = wave < 3 ? 2 : 1;  // 7 row tiles of 32 (196 rows): waves 0-2 take two, wave 3 one
  int abase[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int m = (wave + 4 * t) * 32 + r;
    m = m < 196 ? m : 195;  // clamp the padding rows of the last tile; their results are discarded
    const int win = m >> 2, sub = m & 3;
    const int py = win / 7, px = win - py * 7;
    const int y = 2 * py + (sub >> 1), xx = 2 * px + (sub & 1);
    abase[t] = (y * 18 + xx) * PIX + h * 16;
  }
  const int g = lane >> 4, i16 = lane & 15;
  const int boff = (8 * (g >> 1) + (i16 >> 2)) * WROW + (16 * (g & 1) + 4 * (i16 & 3)) * 2;

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // Software-pipelined k loop: one wave per SIMD means nothing else hides the LDS latency, and issued in program order
  // (read, wait, MFMA) a k-step costs ~420 cycles, 8.8 of the kernel's 11.7 us.  The fragments of k-step ks + PF are
  // fetched before the MFMAs of k-step ks are issued (fully unrolled: the ring indices are compile-time constants).
  // Wave 3 computes a second, clamped row tile like the others (its result is discarded below): no branch in the loop.
  constexpr int PF = 2, RING = PF + 1, KS = 50;
  bf16x8 bq[RING], a0q[RING], a1q[RING];
  auto fetch = [&](const int ks, const int slot) {
    const int tap = ks >> 1, half = ks & 1;
    const int toff = ((tap / 5) * 18 + (tap % 5)) * PIX;
    bq[slot] = tr_frag(wl + (tap * 32 + half * 16) * WROW + boff);
    a0q[slot] = *reinterpret_cast<const bf16x8*>(img + abase[0] + toff + half * 32);
    a1q[slot] = *reinterpret_cast<const bf16x8*>(img + abase[1] + toff + half * 32);
  };
#pragma unroll
  for (int ks = 0; ks < PF; ++ks) fetch(ks, ks % RING);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + PF < KS) fetch(ks + PF, (ks + PF) % RING);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0q[ks % RING], bq[ks % RING], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1q[ks % RING], bq[ks % RING], acc[1], 0, 0, 0);
  }

  // ---- epilogue: bias + ReLU + 2x2 max-pool on register quads (pool-window row order)
  const int col = n0 + r;
  const float bv = bias[col];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < ntile) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int row = (wave + 4 * t) * 32 + 8 * gq + 4 * h;
        float best = acc[t][4 * gq];
        int arg = 0;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
          const float v = acc[t][4 * gq + j];
          if (v > best) { best = v; arg = j; }
        }
        if (row < 196) {
          const long o = ((long)b * 49 + (row >> 2)) * 64 + col;
          out[o] = f2bf(fmaxf(best + bv, 0.f));
          if (amax) amax[o] = (uint8_t)arg;
        }
      }
    }
  }
}

static bool g_attr_set = false;
static char* g_pack = nullptr;  // packed filter copies for callers that pass fp32 weights (the step executor keeps its own)

__global__ __launch_bounds__(256) void conv2_pack_kernel(const float* __restrict__ W, char* __restrict__ Wf, char* __restrict__ Wb) {
  conv2_pack_body(W, Wf, Wb, blockIdx.x, gridDim.x);
}

// Raises the dynamic-LDS limit of the kernels in this file (> 64 KB).  Called once per process before the
// first launch and never from inside a stream capture.
DTF_API int dtf_conv2_lds_init() {
  if (g_attr_set) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_fwd_lds_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, IMG_BYTES + W_BYTES);
  if (e != hipSuccess) return (int)e;
  if (!g_pack && hipMalloc(reinterpret_cast<void**>(&g_pack), CONV2_WF_BYTES + CONV2_WB_BYTES) != hipSuccess) return -12;
  g_attr_set = true;
  return 0;
}

// Wf: packed bf16 filter [2][800][32] (conv2_pack.h)
DTF_API int dtf_conv2_fwd_lds_packed(const void* x, const void* Wf, const float* bias, void* out, void* amax, int B,
                                     hipStream_t st) {
  if (B <= 0 || !Wf) return ERR_SHAPE;
  const int lds = IMG_BYTES + W_BYTES;
  if (!g_attr_set) {
    const int rc = dtf_conv2_lds_init();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(conv2_fwd_lds_kernel, dim3(B, 2), dim3(256), lds, st, (const uint16_t*)x, (const char*)Wf, bias,
                     (uint16_t*)out, (uint8_t*)amax);
  DTF_CHECK_LAUNCH();
  return 0;
}

// fp32 HWIO filter [800][64]: packed into a library-owned scratch first (one extra small launch)
DTF_API int dtf_conv2_fwd_lds(const void* x, const float* W, const float* bias, void* out, void* amax, int B,
                              hipStream_t st) {
  if (B <= 0) return ERR_SHAPE;
  if (!g_attr_set) {
    const int rc = dtf_conv2_lds_init();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(conv2_pack_kernel, dim3(25), dim3(256), 0, st, W, g_pack, g_pack + CONV2_WF_BYTES);
  DTF_CHECK_LAUNCH();
  return dtf_conv2_fwd_lds_packed(x, g_pack, bias, out, amax, B, st);
}

// =====================================================================================================
// Backward, LDS-resident.  Both kernels rebuild the dense conv2 output gradient on the fly from the pooled
// gradient (fc1 bwd-data result), the forward pooled activations (ReLU mask) and the pool argmax, so the
// dense dY tensor and the separate un-pool kernel disappear.
// =====================================================================================================
namespace {
constexpr int DPIX = 144;                    // bytes per dY halo pixel (64 ch bf16 + pad)
constexpr int DIMG_BYTES = 18 * 18 * DPIX;   // 46 656
constexpr int WB_ROW = 128;                  // bwd-data filter row [tap][cin] -> 64 cout bf16, XOR-swizzled 16-B chunks
constexpr int WB_BYTES = 800 * WB_ROW;       // 102 400

// Expand one image's pooled gradient into the zero-initialised dense halo image in LDS.
// task = (window 0..48, 8-channel chunk 0..7); returns the masked gradient values for the bias gradient.
__device__ __forceinline__ void unpool_task_load(const float* __restrict__ dpool, const uint16_t* __restrict__ pooled,
                                                 const uint8_t* __restrict__ amax, long img_off, int task, bool ok,
                                                 f32x4& g0, f32x4& g1, uint4& pv, uint2& av) {
  const long o = img_off + (ok ? task * 8 : 0);
  g0 = *reinterpret_cast<const f32x4*>(dpool + o);
  g1 = *reinterpret_cast<const f32x4*>(dpool + o + 4);
  pv = *reinterpret_cast<const uint4*>(pooled + o);
  av = *reinterpret_cast<const uint2*>(amax + o);
}
__device__ __forceinline__ void unpool_task_store(char* dimg, int task, const f32x4& g0, const f32x4& g1, const uint4& pv,
                                                  const uint2& av, float* gsum /*[8] or null*/) {
  const int win = task >> 3, c0 = (task & 7) * 8;
  const int py = win / 7, px = win - py * 7;
  float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
  const uint32_t pw[4] = {pv.x, pv.y, pv.z, pv.w};
  int a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t bits = (pw[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
    const bool pos = (bits & 0x7fffu) != 0 && (bits & 0x8000u) == 0;
    g[j] = pos ? g[j] : 0.f;
    a[j] = ((j < 4 ? av.x : av.y) >> ((j & 3) * 8)) & 0xff;
    if (gsum) gsum[j] = g[j];
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint32_t ov[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) ov[j >> 1] = pack_bf2(a[j] == s ? g[j] : 0.f, a[j + 1] == s ? g[j + 1] : 0.f);
    const int hy = 2 * py + (s >> 1) + 2, hx = 2 * px + (s & 1) + 2;
    *reinterpret_cast<uint4*>(dimg + (hy * 18 + hx) * DPIX + c0 * 2) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
  }
}
}  // namespace

// ---- conv2 backward-data + un-pool + ReLU mask + bias gradient.   grid (B, 2): row tiles {0..3} / {4..6}.
// dpool2 fp32 [B,49,64]; pool2 bf16 [B,49,64]; amax2 u8 [B,49,64]; W fp32 HWIO [25][32][64];
// pool1 bf16 [B,196,32] (ReLU mask of the layer below); dpool1 fp32 [B,196,32]; db2[64] (atomics).
__global__ __launch_bounds__(256) void conv2_bwd_data_lds_kernel(const float* __restrict__ dpool2,
                                                                 const uint16_t* __restrict__ pool2,
                                                                 const uint8_t* __restrict__ amax2,
                                                                 const char* __restrict__ Wb,
                                                                 const uint16_t* __restrict__ pool1,
                                                                 float* __restrict__ dpool1, float* __restrict__ db2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dimg = smem;
  char* wl = smem + DIMG_BYTES;
  __shared__ float dbs[64];
  const int b = blockIdx.x, half = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // one burst: un-pool inputs (2 tasks / thread) + the filter image
  f32x4 ug0[2], ug1[2];
  uint4 upv[2];
  uint2 uav[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int task = tid + 256 * i;
    unpool_task_load(dpool2, pool2, amax2, (long)b * 3136, task, task < 392, ug0[i], ug1[i], upv[i], uav[i]);
  }
  // the packed, pre-swizzled filter image (conv2_pack.h): 102 400 B = 25 chunks of 16 B per thread, copied as is
  u32x4 wv[25];
#pragma unroll
  for (int j = 0; j < 25; ++j) wv[j] = reinterpret_cast<const u32x4*>(Wb)[tid + 256 * j];
  // zero the halo image while the loads fly (2916 x 16 B)
  if (tid < 64) dbs[tid] = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int c = tid + 256 * i;
    if (c < DIMG_BYTES / 16) *reinterpret_cast<uint4*>(dimg + c * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  float gsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int task = tid + 256 * i;
    if (task < 392) {
      float gs[8];
      unpool_task_store(dimg, task, ug0[i], ug1[i], upv[i], uav[i], gs);
      if (i == half) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gsum[j] = gs[j];
      }
    }
  }
  // Bias gradient: both workgroups of an image un-pool all of it, so each contributes one of the two task rounds.
  // The 8 lanes of a wave that share a channel chunk (lane & 7) are summed with shuffles first: per-task LDS atomics
  // put 8 lanes on every address and cost the row-half-0 workgroups 3.5 us (phase timestamps), a quarter of the kernel.
  if (db2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = gsum[j];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 8) atomicAdd(&dbs[lane * 8 + j], v);
    }
  }
#pragma unroll
  for (int j = 0; j < 25; ++j) *reinterpret_cast<u32x4*>(wl + (tid + 256 * j) * 16) = wv[j];
  __syncthreads();
  if (db2 && tid < 64) atomicAdd(&db2[tid], dbs[tid]);

  const int tile = half * 4 + wave;
  if (tile >= 7) return;
  const int r = lane & 31, h = lane >> 5;
  int m = tile * 32 + r;
  m = m < 196 ? m : 195;
  const int y = m / 14, xx = m - y * 14;
  const int abase = (y * 18 + xx) * DPIX + h * 16;
  // the ReLU mask of the layer below, needed by the epilogue only: in flight during the k loop
  uint16_t mk[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    mk[i] = pool1[((long)b * 196 + (row < 196 ? row : 195)) * 32 + r];
  }
  // software-pipelined k loop (see conv2_fwd_lds_kernel): fragments of k-step ks + PF are fetched before the MFMA of
  // k-step ks is issued; even and odd k-steps accumulate separately so that consecutive MFMAs are independent
  constexpr int PF = 3, RING = PF + 1, KS = 100;
  f32x16 acc2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.f;
  bf16x8 aq[RING], bq[RING];
  auto fetch = [&](const int ks, const int slot) {
    const int tapp = ks >> 2, kk = ks & 3;
    const int toff = ((tapp / 5) * 18 + (tapp % 5)) * DPIX;
    const int wrow = (24 - tapp) * 32 + r;  // flipped tap, this lane's cin
    aq[slot] = *reinterpret_cast<const bf16x8*>(dimg + abase + toff + kk * 32);
    bq[slot] = *reinterpret_cast<const bf16x8*>(wl + wrow * WB_ROW + (((2 * kk + h) ^ (wrow & 7)) << 4));
  };
#pragma unroll
  for (int ks = 0; ks < PF; ++ks) fetch(ks, ks % RING);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + PF < KS) fetch(ks + PF, (ks + PF) % RING);
    acc2[ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks % RING], bq[ks % RING], acc2[ks & 1], 0, 0, 0);
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = acc2[0][i] + acc2[1][i];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (row < 196) {
      const long o = ((long)b * 196 + row) * 32 + r;
      dpool1[o] = ((mk[i] & 0x7fffu) != 0 && (mk[i] & 0x8000u) == 0) ? acc[i] : 0.f;
    }
  }
}

// ---- conv2 filter gradient.   grid (ceil(B / ipw), 5 tap rows), 320 threads: wave = tap column dx.
// pool1 bf16 [B,196,32] (the conv input); dW fp32 [25][32][64] accumulated with atomics.
__device__ __forceinline__ void conv2_dw_body(char* __restrict__ smem, const int bx, const int dy,
                                              const uint16_t* __restrict__ pool1, const float* __restrict__ dpool2,
                                              const uint16_t* __restrict__ pool2, const uint8_t* __restrict__ amax2,
                                              float* __restrict__ dW, int B, int ipw) {
  char* ximg = smem;               // 18x18 halo x 32 ch, 80-B pixels
  char* dimg = smem + IMG_BYTES;   // 18x18 halo x 64 ch, 144-B pixels
  const int tid = threadIdx.x, lane = tid & 63, dx = tid >> 6;
  const int b0 = bx * ipw;
  const int nimg = min(ipw, B - b0);

  // per-lane LDS offsets of the transposing reads, identical for every image (13 k-steps x 2 reads)
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = (i16 & 3) * 4;
  const int mcol = 16 * (g & 1) + p4;  // A: cin column / B: cout column (before the n-tile offset)
  // (byte offsets < 64 K, two per register: with 52 full registers of offsets the launch exceeds 128 VGPRs, and a
  // 5-wave workgroup above 128 VGPRs gets a CU to itself -- the optimizer workgroups of the fused launch then start
  // only when a filter-gradient workgroup has finished, 9.7 us into the launch on average)
  uint32_t xo2[13], do2[13];
#pragma unroll
  for (int s = 0; s < 26; ++s) {
    const int pos = (s >> 1) * 16 + 8 * (g >> 1) + (s & 1) * 4 + q;
    const bool ok = pos < 196;
    const int y = ok ? pos / 14 : 0, xx = ok ? pos - (pos / 14) * 14 : 0;
    const uint32_t xo = ((y + dy) * 18 + xx + dx) * PIX + mcol * 2;
    const uint32_t dofs = (ok ? ((y + 2) * 18 + xx + 2) * DPIX : 0) + mcol * 2;  // padded k -> halo pixel 0 (zeros)
    if (s & 1) { xo2[s >> 1] |= xo << 16; do2[s >> 1] |= dofs << 16; } else { xo2[s >> 1] = xo; do2[s >> 1] = dofs; }
  }
  // zero both halo images once: interiors are fully rewritten per image, halos stay zero
  for (int c = tid; c < (IMG_BYTES + DIMG_BYTES) / 16; c += 320) *reinterpret_cast<uint4*>(smem + c * 16) = make_uint4(0, 0, 0, 0);

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int im = 0; im < nimg; ++im) {
    const int b = b0 + im;
    // loads for this image (X: 784 chunks -> 3 / thread; un-pool: 392 tasks -> 2 / thread)
    u32x4 xv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = tid + 320 * i;
      xv[i] = *reinterpret_cast<const u32x4*>(pool1 + (long)b * 6272 + (c < 784 ? c * 8 : 0));
    }
    f32x4 ug0[2], ug1[2];
    uint4 upv[2];
    uint2 uav[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int task = tid + 320 * i;
      unpool_task_load(dpool2, pool2, amax2, (long)b * 3136, task, task < 392, ug0[i], ug1[i], upv[i], uav[i]);
    }
    __syncthreads();  // previous image fully consumed (and the zero fill done)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = tid + 320 * i;
      if (c < 784) {
        const int pix = c >> 2, y = pix / 14, xx = pix - y * 14;
        *reinterpret_cast<u32x4*>(ximg + ((y + 2) * 18 + xx + 2) * PIX + (c & 3) * 16) = xv[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int task = tid + 320 * i;
      if (task < 392) unpool_task_store(dimg, task, ug0[i], ug1[i], upv[i], uav[i], nullptr);
    }
    __syncthreads();
    // software-pipelined (see conv2_fwd_lds_kernel): the six transposing reads of k-step ks + PF before the MFMAs of ks
    constexpr int PF = 2, RING = PF + 1, KS = 13;
    bf16x8 aq[RING], b0q[RING], b1q[RING];
    auto tr8 = [&](const char* lo_p, const char* hi_p) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lo_p));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(hi_p));
      const s16x8 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v8);
    };
    auto fetch = [&](const int ks, const int slot) {
      // (volatile: unpacked at the point of use; hoisted out of the image loop the 52 offsets would be registers again)
      uint32_t x0, x1, d0, d1;
      asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(x0) : "v"(xo2[ks]));
      asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(x1) : "v"(xo2[ks]));
      asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(d0) : "v"(do2[ks]));
      asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(d1) : "v"(do2[ks]));
      aq[slot] = tr8(ximg + x0, ximg + x1);
      b0q[slot] = tr8(dimg + d0, dimg + d1);
      b1q[slot] = tr8(dimg + d0 + 64, dimg + d1 + 64);
    };
#pragma unroll
    for (int ks = 0; ks < PF; ++ks) fetch(ks, ks % RING);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + PF < KS) fetch(ks + PF, (ks + PF) % RING);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks % RING], b0q[ks % RING], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks % RING], b1q[ks % RING], acc[1], 0, 0, 0);
    }
  }
  // dW[tap][cin = row][cout = col]: a 32-lane half adds 128 contiguous bytes per instruction
  const int tap = dy * 5 + dx, r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      atomicAdd(&dW[((long)tap * 32 + row) * 64 + nt * 32 + r], acc[nt][i]);
    }
}

__global__ __launch_bounds__(320) void conv2_dw_lds_kernel(const uint16_t* __restrict__ pool1,
                                                           const float* __restrict__ dpool2,
                                                           const uint16_t* __restrict__ pool2,
                                                           const uint8_t* __restrict__ amax2, float* __restrict__ dW,
                                                           int B, int ipw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  conv2_dw_body(smem, blockIdx.x, blockIdx.y, pool1, dpool2, pool2, amax2, dW, B, ipw);
}

// conv2 dW (workgroups [0, n2), id -> (image group id % gx, tap row id / gx)) and conv1 dW/db (the rest) in one launch.
// (+ optionally, trailing workgroups that apply Adam to a parameter range whose gradient is already final — the fc1
// weights, 98 % of the optimizer's bytes: the bandwidth-bound update then runs next to these latency-bound kernels
// instead of after them.  Horizontal fusion, because a forked stream costs more than it hides; see engine.hip.)
struct ConvDwDuo {
  const uint16_t* pool1; const float* dpool2; const uint16_t* pool2; const uint8_t* amax2; float* dW2;
  int B, ipw, gx, n2;
  const float* images; const int* perm; const int* step_ptr; int spe;
  const float* dpool1; const uint8_t* amax1; float* dW1; float* db1;
  int n1;
  OptimParams adam;  // adam.n == 0: no optimizer workgroups
  int nadam;
};
constexpr int DUO_LDS = (IMG_BYTES + DIMG_BYTES) > CONV1_DW_LDS_FLOATS * 4 ? (IMG_BYTES + DIMG_BYTES) : CONV1_DW_LDS_FLOATS * 4;

__global__ __launch_bounds__(320, 4) void conv_dw_duo_kernel(const ConvDwDuo a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int id = blockIdx.x;
  if (id < a.n2)
    conv2_dw_body(smem, id % a.gx, id / a.gx, a.pool1, a.dpool2, a.pool2, a.amax2, a.dW2, a.B, a.ipw);
  else if (id >= a.n2 + a.n1)
    adam_tf_body(a.adam, id - a.n2 - a.n1, a.nadam, 320);
  else
    conv1_dw_body(reinterpret_cast<float*>(smem), id - a.n2, a.n1, a.images, a.perm, a.step_ptr, a.spe, a.B, a.dpool1,
                  a.amax1, a.dW1, a.db1);
}


static bool g_bwd_attr_set = false;
DTF_API int dtf_conv2_bwd_lds_init() {
  if (g_bwd_attr_set) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_bwd_data_lds_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, DIMG_BYTES + WB_BYTES);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_dw_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          IMG_BYTES + DIMG_BYTES);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dw_duo_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          DUO_LDS);
  if (e != hipSuccess) return (int)e;
  g_bwd_attr_set = true;
  return 0;
}

// Fused conv2 backward for the MNIST geometry (14x14, 32 -> 64 channels, 5x5 SAME, 2x2 pool behind it).
// The two kernels are independent of each other (each rebuilds dY from the pooled gradient), so the step
// executor may issue them on different streams.
// Wb: packed, swizzled bf16 filter image [800][64] (conv2_pack.h)
DTF_API int dtf_conv2_bwd_data_lds_packed(const float* dpool2, const void* pool2, const void* amax2, const void* Wb,
                                          const void* pool1, float* dpool1, float* db2, int B, hipStream_t st) {
  if (B <= 0 || !Wb) return ERR_SHAPE;
  if (!g_bwd_attr_set) {
    const int rc = dtf_conv2_bwd_lds_init();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(conv2_bwd_data_lds_kernel, dim3(B, 2), dim3(256), DIMG_BYTES + WB_BYTES, st, dpool2,
                     (const uint16_t*)pool2, (const uint8_t*)amax2, (const char*)Wb, (const uint16_t*)pool1, dpool1, db2);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_conv2_bwd_data_lds(const float* dpool2, const void* pool2, const void* amax2, const float* W,
                                   const void* pool1, float* dpool1, float* db2, int B, hipStream_t st) {
  if (B <= 0) return ERR_SHAPE;
  if (!g_bwd_attr_set || !g_attr_set) {
    const int rc = dtf_conv2_bwd_lds_init() || dtf_conv2_lds_init();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(conv2_pack_kernel, dim3(25), dim3(256), 0, st, W, g_pack, g_pack + CONV2_WF_BYTES);
  DTF_CHECK_LAUNCH();
  return dtf_conv2_bwd_data_lds_packed(dpool2, pool2, amax2, g_pack + CONV2_WF_BYTES, pool1, dpool1, db2, B, st);
}

DTF_API int dtf_conv2_dw_lds(const float* dpool2, const void* pool2, const void* amax2, const void* pool1, float* dW,
                             int B, int ipw, hipStream_t st) {
  if (B <= 0 || ipw <= 0) return ERR_SHAPE;
  if (!g_bwd_attr_set) {
    const int rc = dtf_conv2_bwd_lds_init();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(conv2_dw_lds_kernel, dim3(dtf_cdiv(B, ipw), 5), dim3(320), IMG_BYTES + DIMG_BYTES, st,
                     (const uint16_t*)pool1, dpool2, (const uint16_t*)pool2, (const uint8_t*)amax2, dW, B, ipw);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Both filter gradients of the conv stack in one launch (see ConvDwDuo): conv2's from (pool1, dpool2, pool2, amax2),
// conv1's from (images[perm], dpool1, amax1).  dW2 / dW1 / db1 are accumulated with atomics (pre-zeroed).
DTF_API int dtf_conv_dw_duo(const float* dpool2, const void* pool2, const void* amax2, const void* pool1, float* dW2, int B,
                            int ipw, const float* images, const int* perm, const int* step_ptr, int spe,
                            const float* dpool1, const void* amax1, float* dW1, float* db1, const OptimParams* adam,
                            hipStream_t st) {
  if (B <= 0 || ipw <= 0 || (perm && spe <= 0)) return ERR_SHAPE;
  if (adam && (adam->n <= 0 || adam->n % 4 || adam->zero_from % 4 || !adam->step)) return ERR_SHAPE;
  if (!g_bwd_attr_set) {
    const int rc = dtf_conv2_bwd_lds_init();
    if (rc) return rc;
  }
  ConvDwDuo a;
  a.pool1 = (const uint16_t*)pool1; a.dpool2 = dpool2; a.pool2 = (const uint16_t*)pool2; a.amax2 = (const uint8_t*)amax2;
  a.dW2 = dW2; a.B = B; a.ipw = ipw; a.gx = dtf_cdiv(B, ipw); a.n2 = a.gx * 5;
  a.images = images; a.perm = perm; a.step_ptr = step_ptr; a.spe = spe;
  a.dpool1 = dpool1; a.amax1 = (const uint8_t*)amax1; a.dW1 = dW1; a.db1 = db1;
  a.n1 = B < 128 ? B : 128;
  a.adam = OptimParams{};
  a.nadam = 0;
  if (adam) {
    a.adam = *adam;
    a.adam.ticket = nullptr;  // (the step is advanced by the optimizer launch that covers the rest of the arena)
    // two workgroups of this launch fit on a CU (72 KB of LDS each): fill what the filter-gradient workgroups leave
    // (measured at batch 100, 4 images per conv2 workgroup: 150 -> 117.7, 287 = fill -> 114.7, 400 -> 115.4, 800 -> 115.9 us per step)
    a.nadam = 512 - (a.n2 + a.n1) > 64 ? 512 - (a.n2 + a.n1) : 64;
  }
  hipLaunchKernelGGL(conv_dw_duo_kernel, dim3(a.n2 + a.n1 + a.nadam), dim3(320), DUO_LDS, st, a);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_conv2_bwd_lds(const float* dpool2, const void* pool2, const void* amax2, const float* W, const void* pool1,
                              float* dpool1, float* dW, float* db2, int B, int ipw, hipStream_t st) {
  const int rc = dtf_conv2_bwd_data_lds(dpool2, pool2, amax2, W, pool1, dpool1, db2, B, st);
  if (rc) return rc;
  return dtf_conv2_dw_lds(dpool2, pool2, amax2, pool1, dW, B, ipw, st);
}
