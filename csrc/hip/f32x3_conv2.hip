// fp32 conv2 (14x14x32 -> 14x14x64, 5x5 SAME, 2x2 pool) on the bf16 matrix cores: every operand is three bf16 planes and
// every product six bf16 MFMAs (bf16x3.h), 6/16 of the exact-fp32 MFMA's time at fp32 accuracy — the same tests
// (tests/test_f32_gpu.py, float64 oracle, 1e-5) hold.  Reference semantics: tf.nn.conv2d on float32 + bias + ReLU + 2x2
// max-pool, /root/reference/demo1/train.py:40-45,84-85.
//
// Both kernels keep the decomposition of their exact-fp32 counterparts (f32_conv2.hip): one workgroup = 80 output rows
// (5 MFMA row tiles) x all output columns, 245 workgroups at batch 100, 8 waves that split K and meet in LDS at the end.
//  * the fp32 activations are split into planes ONCE, while the halo'd image is staged into LDS (three images of 16-byte
//    fragments, one per plane); only the halo rows the 80 output rows touch are staged (<= 16), so the three planes fit;
//  * the filter's planes are written once per step by spare workgroups of the conv1 forward launch (conv2_pack3.h) and
//    stream from L2 straight into B fragments, one k-unit ahead;
//  * a k-unit is 32 consecutive k values = one v_mfma_f32_16x16x32_bf16 per plane pair: a whole tap in the forward (32 input
//    channels), half a tap in the data gradient (64 output channels); per unit a wave reads 15 A fragments and issues 60
//    MFMAs on its 10 accumulator tiles.
#include "common.h"
#include "bf16x3.h"
#include "conv2_pack3.h"
#include "conv1_dw_body.h"
#include "adam_body.h"
#include "dtf_api.h"

namespace {
constexpr int ERR_SHAPE = -22;

// ---- forward: plane image = 16 halo rows x 18 pixels x (32 bf16 + 16 B pad).  A fragments are ds_read_b128 of the 16
// pixels of 4 pooling windows = 8 columns x 2 rows: 5 sixteen-byte units per pixel (odd: 8 columns, 8 different units mod
// 16), and the second row must add 8 units: 1664 / 16 = 104 = 8 mod 16.
constexpr int F3_PIX = 80, F3_ROW = 1664, F3_ROWS = 16, F3_PLANE = F3_ROWS * F3_ROW;  // 26 624 B per plane
constexpr int F3_WIN = 20, F3_MT = F3_WIN / 4;
constexpr int F3_RED = 8 * F3_MT * 2 * 256 * 4;  // [wave][tile][lane][4] floats: 81 920 B
constexpr int F3_LDS = 3 * F3_PLANE > F3_RED ? 3 * F3_PLANE : F3_RED;

// ---- data gradient: plane image = 15 halo rows x 18 pixels x (64 bf16 + 16 B pad).  A fragments are the 16 consecutive
// positions of a row tile: 9 units per pixel (consecutive pixels, consecutive odd multiples), and a tile that runs over a
// row end continues 4 halo pixels further, which the row pitch turns into "the next pixel": pitch / 16 = 14 * 9 mod 16 =
// 14 -> 174 units.
constexpr int D3_PIX = 144, D3_ROW = 2784, D3_ROWS = 15, D3_PLANE = D3_ROWS * D3_ROW;  // 41 760 B per plane
constexpr int D3_POS = 80, D3_MT = D3_POS / 16;
constexpr int D3_RED = 8 * D3_MT * 2 * 256 * 4;
constexpr int D3_LDS = 3 * D3_PLANE > D3_RED ? 3 * D3_PLANE : D3_RED;  // 125 280 B: one workgroup per CU

__device__ __forceinline__ void store_planes8(char* dst, const int plane_bytes, const Bf3Quad& s) {
  *reinterpret_cast<uint2*>(dst) = s.p0;
  *reinterpret_cast<uint2*>(dst + plane_bytes) = s.p1;
  *reinterpret_cast<uint2*>(dst + 2 * plane_bytes) = s.p2;
}

// the six plane products of 5 x 2 tiles, plane pair outermost: ten independent MFMAs between two that share an accumulator
#define X3_PROD(PA, PB)                                                                  \
  _Pragma("unroll") for (int mt = 0; mt < 5; ++mt) {                                     \
    acc[mt][0] = mfma16bf(a[mt][PA], bcur[PB][0], acc[mt][0]);                           \
    acc[mt][1] = mfma16bf(a[mt][PA], bcur[PB][1], acc[mt][1]);                           \
  }
#define X3_ALL_PRODUCTS() do { X3_PROD(2, 0) X3_PROD(0, 2) X3_PROD(1, 1) X3_PROD(1, 0) X3_PROD(0, 1) X3_PROD(0, 0) } while (0)
}  // namespace

// ------------------------------------------------------------------ forward + bias + ReLU + 2x2 max-pool
// One workgroup = 20 consecutive pooling windows of the batch's 49 B; rows in pool-window order (m = 4 window + sub), so a
// lane's four accumulator registers are one pooling window.  Wave = (output-channel half nh, tap group kq): taps kq, kq + 4,
// ... (7 or 6 of the 25; the group is rotated with nh so that every SIMD gets 13 or 12).
// grid ceil(49 B / 20); 512 threads.  x fp32 [B,14,14,32]; Wf3 (conv2_pack3.h); out fp32 [B,49,64]; amax uint8 [B,49,64]
__global__ __launch_bounds__(512) void conv2_fwd_x3_kernel(const float* __restrict__ x, const char* __restrict__ Wf3,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           uint8_t* __restrict__ amax, const int B) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nh = wave >> 2, kq = (wave + nh) & 3;
  const int nwin = 49 * B;
  const int gw0 = blockIdx.x * F3_WIN;
  const int b0 = gw0 / 49;
  const int glast = min(gw0 + F3_WIN, nwin) - 1;
  const bool two = glast / 49 != b0;  // (workgroup-uniform)
  // staged halo rows (halo row = image row + 2; window row py reads halo rows 2 py .. 2 py + 5): slot 0 = image b0, window
  // rows [pylo0, pyhi0] at LDS rows 0 .., slot 1 = image b0 + 1, window rows [0, pyhi1] behind them; 20 windows span at
  // most 4 window rows over both slots = at most 16 halo rows
  const int pylo0 = (gw0 - b0 * 49) / 7;
  const int pyhi0 = two ? 6 : (glast - b0 * 49) / 7;
  const int hlo0 = 2 * pylo0;
  const int nrow0 = 2 * (pyhi0 - pylo0) + 6;
  const int nrow1 = two ? 2 * ((glast - (b0 + 1) * 49) / 7) + 6 : 0;

  // B fragments of tap t: plane p, tile nt: Wf3[p][t][32 nh + 16 nt + r][8 q .. 8 q + 7]
  const char* wb = Wf3 + ((long)(32 * nh + r) * 32 + 8 * q) * 2;
  bf16x8 bcur[3][2], bnxt[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bcur[p][nt] = *reinterpret_cast<const bf16x8*>(wb + p * CONV2_W3_PLANE + ((long)kq * 64 + 16 * nt) * 64);

  // ---- the staged rows: (row, halo column, 4-channel part) chunks, every load issued before the first split
  {
    const int nchunk = (nrow0 + nrow1) * 144;
    f32x4 iv[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + 512 * i;
      const int row = c / 144, rem = c - row * 144;
      const int hx = rem >> 3, part = rem & 7;
      const int slot = row >= nrow0 ? 1 : 0;
      const int y = (slot ? row - nrow0 : hlo0 + row) - 2, xx = hx - 2;
      const bool ok = c < nchunk && (unsigned)y < 14u && (unsigned)xx < 14u;
      iv[i] = *reinterpret_cast<const f32x4*>(x + (ok ? (((long)(b0 + slot) * 14 + y) * 14 + xx) * 32 + part * 4 : 0L));
      if (!ok) iv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = tid + 512 * i;
      const int row = c / 144, rem = c - row * 144;
      if (c < nchunk) store_planes8(smem3 + row * F3_ROW + (rem >> 3) * F3_PIX + (rem & 7) * 8, F3_PLANE, bf3_split4(iv[i]));
    }
  }
  int abase[F3_MT];
#pragma unroll
  for (int mt = 0; mt < F3_MT; ++mt) {
    int gw = gw0 + mt * 4 + (r >> 2);
    gw = gw < nwin ? gw : nwin - 1;  // padding windows of the last workgroup; their results are discarded
    const int slot = gw / 49 - b0, win = gw - (b0 + slot) * 49, sub = r & 3;
    const int py = win / 7, px = win - py * 7;
    const int lrow = (slot ? nrow0 : -hlo0) + 2 * py + (sub >> 1);
    abase[mt] = lrow * F3_ROW + (2 * px + (sub & 1)) * F3_PIX + 16 * q;
  }
  f32x4 acc[F3_MT][2];
#pragma unroll
  for (int mt = 0; mt < F3_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int ntap = kq == 0 ? 7 : 6;
#pragma unroll 1
  for (int i = 0; i < ntap; ++i) {
    const int t = kq + 4 * i;
    const int tn = t + 4 < 25 ? t + 4 : t;  // (the last prefetch is a harmless repeat)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bnxt[p][nt] = *reinterpret_cast<const bf16x8*>(wb + p * CONV2_W3_PLANE + ((long)tn * 64 + 16 * nt) * 64);
    const int dy = t / 5, dx = t - dy * 5;
    const char* ab = smem3 + dy * F3_ROW + dx * F3_PIX;
    bf16x8 a[F3_MT][3];
    // (plane 2 first: the products that need it come first, and LDS returns in order)
#pragma unroll
    for (int mt = 0; mt < F3_MT; ++mt) a[mt][2] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + 2 * F3_PLANE);
#pragma unroll
    for (int mt = 0; mt < F3_MT; ++mt) a[mt][0] = *reinterpret_cast<const bf16x8*>(ab + abase[mt]);
#pragma unroll
    for (int mt = 0; mt < F3_MT; ++mt) a[mt][1] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + F3_PLANE);
    // (pinned: left to itself the compiler sinks the filter prefetch to a few MFMAs before its use and reads the A fragments
    // two at a time just in time, so every unit starts with a chain of exposed round trips)
    __builtin_amdgcn_sched_barrier(0);
    X3_ALL_PRODUCTS();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bcur[p][nt] = bnxt[p][nt];
  }

  // ---- combine the four K slices of each channel half through LDS (the image is dead), then bias + ReLU + pool on register
  // quads; tile id = (nh * F3_MT + mt) * 2 + nt, finalised by wave (id % 8)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem3);
  constexpr int TPW = F3_MT * 2;  // tiles per wave
#pragma unroll
  for (int mt = 0; mt < F3_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<f32x4*>(red + ((wave * TPW + mt * 2 + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();
  for (int id = wave; id < 2 * TPW; id += 8) {
    const int h = id / TPW, t = id - h * TPW;
    f32x4 s = *reinterpret_cast<const f32x4*>(red + (((4 * h) * TPW + t) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + (((4 * h + w) * TPW + t) * 64 + lane) * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int mt = t >> 1, nt = t & 1;
    const int gw = gw0 + mt * 4 + q, col = 32 * h + nt * 16 + r;
    float best = s[0];
    int arg = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (s[j] > best) { best = s[j]; arg = j; }
    if (gw < nwin) {
      const long o = (long)gw * 64 + col;
      out[o] = fmaxf(best + bias[col], 0.f);
      if (amax) amax[o] = (uint8_t)arg;
    }
  }
}

// ------------------------------------------------------------------ data gradient + un-pool + ReLU mask
// One workgroup = 80 consecutive input positions of the batch's 196 B x all 32 input channels.  A k-unit is (tap, half of
// the 64 output channels): 50 units, wave w takes units w, w + 8, ... (7 or 6).  Only the halo rows the 80 positions need
// (<= 15) are un-pooled into the three plane images.
// grid ceil(196 B / 80); 512 threads.  dpool2 / pool2 fp32 [B,49,64]; amax2 u8 [B,49,64]; Wb3 (conv2_pack3.h); pool1 fp32
// [B,196,32] (ReLU mask of the layer below); dpool1 fp32 [B,196,32]
__global__ __launch_bounds__(512) void conv2_bwd_data_x3_kernel(const float* __restrict__ dpool2, const float* __restrict__ pool2,
                                                                const uint8_t* __restrict__ amax2, const char* __restrict__ Wb3,
                                                                const float* __restrict__ pool1, float* __restrict__ dpool1,
                                                                const int B) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int npos = 196 * B;
  const int P0 = blockIdx.x * D3_POS;
  const int b0 = P0 / 196;
  const int plast = min(P0 + D3_POS, npos) - 1;
  const bool two = plast / 196 != b0;  // (workgroup-uniform, like everything up to the task decode)
  // slot 0 = image b0, output rows [ylo0, yhi0]; slot 1 = image b0 + 1, output rows [0, yhi1].  Output row y reads halo
  // rows y .. y + 4: slot 0 stages halo rows [ylo0, yhi0 + 4] at LDS rows 0 .., slot 1 halo rows [0, yhi1 + 4] behind them.
  const int ylo0 = (P0 - b0 * 196) / 14;
  const int yhi0 = two ? 13 : (plast - b0 * 196) / 14;
  const int yhi1 = two ? (plast - (b0 + 1) * 196) / 14 : -5;
  const int nrow0 = yhi0 - ylo0 + 5, nrow1 = yhi1 + 5;
  const int py0lo = max(ylo0 - 2, 0) >> 1, py0hi = min(yhi0 + 2, 13) >> 1;
  const int nwr0 = py0hi - py0lo + 1;
  const int nwr1 = two ? (min(yhi1 + 2, 13) >> 1) + 1 : 0;
  const int ntask = (nwr0 + nwr1) * 112;  // (window row, px 0..6, 4-channel group 0..15); <= 6 window rows in all

  // B fragments of unit (flipped tap t', half h): plane p, tile nt: Wb3[p][24 - t'][16 nt + r][32 h + 8 q .. + 7]
  const char* wb = Wb3 + ((long)r * 64 + 8 * q) * 2;
  auto bfrag = [&](const int p, const int nt, const int u) {
    return *reinterpret_cast<const bf16x8*>(wb + p * CONV2_W3_PLANE + ((long)(24 - (u >> 1)) * 32 + 16 * nt) * 128 + (u & 1) * 64);
  };
  bf16x8 bcur[3][2], bnxt[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bcur[p][nt] = bfrag(p, nt, wave);

  f32x4 tg[2], tp[2];
  uint32_t tav[2];
  int trow[2], tcol[2], tslot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int task = tid + 512 * i;
    const int t = task < ntask ? task : 0;
    const int wr = t / 112, rem = t - wr * 112;
    const int slot = wr >= nwr0 ? 1 : 0;
    const int py = slot ? wr - nwr0 : py0lo + wr;
    const int px = rem >> 4;
    tslot[i] = slot; trow[i] = py; tcol[i] = px;
    const long off = ((long)(b0 + slot) * 49 + py * 7 + px) * 64 + (rem & 15) * 4;
    tg[i] = *reinterpret_cast<const f32x4*>(dpool2 + off);
    tp[i] = *reinterpret_cast<const f32x4*>(pool2 + off);
    tav[i] = *reinterpret_cast<const uint32_t*>(amax2 + off);
  }
  {
    const int n16 = (nrow0 + nrow1) * (D3_ROW / 16);
    for (int c = tid; c < n16; c += 512)
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(smem3 + p * D3_PLANE + c * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int task = tid + 512 * i;
    if (task < ntask) {
#pragma unroll
      for (int j = 0; j < 4; ++j) tg[i][j] = tp[i][j] > 0.f ? tg[i][j] : 0.f;
      const int slot = tslot[i];
      const int hlo = slot ? 0 : ylo0, hhi = slot ? yhi1 + 4 : yhi0 + 4, rbase = slot ? nrow0 : -ylo0;
      char* dst = smem3 + ((task - (task / 112) * 112) & 15) * 8;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int hy = 2 * trow[i] + (sp >> 1) + 2;
        if (hy >= hlo && hy <= hhi) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ((tav[i] >> (8 * j)) & 0xffu) == (uint32_t)sp ? tg[i][j] : 0.f;
          store_planes8(dst + (rbase + hy) * D3_ROW + (2 * tcol[i] + (sp & 1) + 2) * D3_PIX, D3_PLANE, bf3_split4(o));
        }
      }
    }
  }

  int abase[D3_MT];
#pragma unroll
  for (int mt = 0; mt < D3_MT; ++mt) {
    const int p = min(P0 + mt * 16 + r, plast);  // (padding rows of the last workgroup repeat its last position)
    const int slot = p >= (b0 + 1) * 196 ? 1 : 0;
    const int pos = p - (b0 + slot) * 196;
    const int y = pos / 14, xx = pos - y * 14;
    abase[mt] = ((slot ? nrow0 : -ylo0) + y) * D3_ROW + xx * D3_PIX + 16 * q;
  }
  // ReLU mask of the layer below for the (at most two) tiles this wave finalises, fetched ahead of the k loop
  float pm[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = wave + 8 * u;
    const int mt = t >> 1, nt = t & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = P0 + mt * 16 + 4 * q + i;
      const bool ok = t < 2 * D3_MT && p < npos;
      const float v = pool1[ok ? (long)p * 32 + nt * 16 + r : 0L];
      pm[u][i] = ok ? v : 0.f;
    }
  }
  f32x4 acc[D3_MT][2];
#pragma unroll
  for (int mt = 0; mt < D3_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int nunit = wave < 2 ? 7 : 6;
#pragma unroll 1
  for (int i = 0; i < nunit; ++i) {
    const int u = wave + 8 * i;
    const int un = u + 8 < 50 ? u + 8 : u;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bnxt[p][nt] = bfrag(p, nt, un);
    const int tap = u >> 1;
    const int dy = tap / 5, dx = tap - dy * 5;
    const char* ab = smem3 + dy * D3_ROW + dx * D3_PIX + (u & 1) * 64;
    bf16x8 a[D3_MT][3];
#pragma unroll
    for (int mt = 0; mt < D3_MT; ++mt) a[mt][2] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + 2 * D3_PLANE);
#pragma unroll
    for (int mt = 0; mt < D3_MT; ++mt) a[mt][0] = *reinterpret_cast<const bf16x8*>(ab + abase[mt]);
#pragma unroll
    for (int mt = 0; mt < D3_MT; ++mt) a[mt][1] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + D3_PLANE);
    __builtin_amdgcn_sched_barrier(0);
    X3_ALL_PRODUCTS();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bcur[p][nt] = bnxt[p][nt];
  }

  // ---- combine the eight K slices through LDS (the staged rows are dead); tile id = mt * 2 + nt, finalised by wave id % 8
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem3);
  constexpr int TPW = D3_MT * 2;
#pragma unroll
  for (int mt = 0; mt < D3_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<f32x4*>(red + ((wave * TPW + mt * 2 + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = wave + 8 * u;
    if (t < TPW) {
      f32x4 s = *reinterpret_cast<const f32x4*>(red + (t * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((w * TPW + t) * 64 + lane) * 4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
      const int mt = t >> 1, nt = t & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = P0 + mt * 16 + 4 * q + i;
        if (p < npos) dpool1[(long)p * 32 + nt * 16 + r] = pm[u][i] > 0.f ? s[i] : 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------ filter gradient (+ bias gradient)
// dW[tap][cin][cout] = sum over images and positions of X_halo[pos + tap][cin] * dY[pos][cout]: K runs over positions.
//  * The k index is the pixel index of an 18-wide, 14-row grid (dY is zero in columns 14..17 and behind row 13): the A
//    operand of tap (dy, dx) is then the staged image shifted by dx pixels, whatever the position -- no per-lane offset
//    tables; 8 k-steps of 32 per image for 196 positions.  X is staged as it comes (pixel-major, three planes) and read with
//    the transposing LDS load.
//  * dY never exists un-pooled: LDS holds, per pooling window and output channel, the three planes of the masked pooled
//    gradient and the argmax (8 bytes; 72 windows of an 8 x 9 window grid whose last row / two last columns are zero).  A
//    lane's B fragment is 8 consecutive grid positions of its channel = four windows x the two positions of one window row:
//    one ds_read_b64 per window, and a dword of the fragment is the window's plane value in the low or the high half, or
//    zero, by the argmax.  18 KB instead of the 98 KB of three un-pooled plane images.
// One workgroup = (4 images, half of the output channels, filter row dy) like the exact-fp32 kernel: 250 workgroups at batch
// 100; 8 waves = (k group kq, input-channel tile mt, output-channel tile nt), accumulators [dx]; four k-steps per image and
// wave (20 accumulator registers and one B tile per wave keep the launch at 128 VGPRs).  68 KB of LDS: two workgroups of the
// launch fit on a CU, so the optimizer workgroups of the fused launch are resident from the start.
namespace {
constexpr int DW3_IPW = 4;
constexpr int DW3_XPIX = 64, DW3_XN = 264;     // X plane: 14 staged rows x 18 pixels (+ 12: the largest shifted read) x 32 cin
constexpr int DW3_XPLANE = DW3_XN * DW3_XPIX;  // 16 896 B
constexpr int DW3_PW = 72 * 32 * 8;            // pooled dY of one channel half: [window 8 x 9][cout 32] x 8 B = 18 432 B
constexpr int DW3_IMG = 3 * DW3_XPLANE + DW3_PW;        // 69 120 B
constexpr int DW3_RED = 8 * 5 * 256 * 4;                // [wave][dx][lane][4] floats: 40 960 B
constexpr int DW3_LDS = DW3_IMG + 128;                  // (the bias-gradient scratch sits behind the images)
static_assert(DW3_RED <= DW3_IMG, "LDS carve-up");

typedef short s16x4_x3 __attribute__((ext_vector_type(4)));
typedef short s16x8_x3 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_x3 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr8_x3(const char* p, const int pitch) {
  const s16x4_x3 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_x3*)(p));
  const s16x4_x3 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_x3*)(p + 4 * pitch));
  const s16x8_x3 v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v8);
}

// id -> (image quad id / 10, output-channel half (id / 5) & 1, filter row id % 5); 512 threads
__device__ __forceinline__ void conv2_dw_x3_body(char* __restrict__ smem, const int id, const float* __restrict__ pool1,
                                                 const float* __restrict__ dpool2, const float* __restrict__ pool2,
                                                 const uint8_t* __restrict__ amax2, float* __restrict__ dW,
                                                 float* __restrict__ db2, const int B) {
  char* xl = smem;
  char* pw = smem + 3 * DW3_XPLANE;
  float* bsum = reinterpret_cast<float*>(smem + DW3_IMG);
  const int dy = id % 5, c = (id / 5) & 1, g = id / 10;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int kq = wave & 1, mt = (wave >> 1) & 1, nt = wave >> 2;
  const int b0 = DW3_IPW * g;
  const int nimg = min(DW3_IPW, B - b0);
  const bool bias_role = db2 && dy == 0;

  // zero both images once: the interiors are rewritten per image, halo pixels / padding windows stay zero
  for (int i = tid; i < DW3_IMG / 16; i += 512) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);
  if (tid < 32) bsum[tid] = 0.f;

  // X: staged row sy (0..13) = image row sy + dy - 2; chunk = (sy, x 0..13, 4-channel part 0..7): 1568 chunks
  f32x4 xv[4];
  f32x4 tg, tp;
  uint32_t tav;
  // pooled-gradient task (threads < 392): window tid >> 3, 4-channel group tid & 7 of this half
  const int twin = tid < 392 ? tid >> 3 : 0, tcg = tid & 7;
  // (32-bit element offsets from the image's uniform base: 64-bit per-thread addresses held across the image loop spill)
  int xoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = tid + 512 * i;
    const int pix = ch >> 3, sy = pix / 14, xx = pix - sy * 14;
    const int y = sy + dy - 2;
    xoff[i] = ch < 1568 && (unsigned)y < 14u ? (y * 14 + xx) * 32 + (ch & 7) * 4 : -1;
  }
  const int toff = twin * 64 + c * 32 + tcg * 4;
  auto load = [&](const int b) {
    const float* xb = pool1 + (long)b * 6272;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[i] = *reinterpret_cast<const f32x4*>(xb + (xoff[i] >= 0 ? xoff[i] : 0));
      if (xoff[i] < 0) xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    tg = *reinterpret_cast<const f32x4*>(dpool2 + (long)b * 3136 + toff);
    tp = *reinterpret_cast<const f32x4*>(pool2 + (long)b * 3136 + toff);
    tav = *reinterpret_cast<const uint32_t*>(amax2 + (long)b * 3136 + toff);
  };
  f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = tid + 512 * i;
      const int pix = ch >> 3, sy = pix / 14, xx = pix - sy * 14;
      // (rows outside the image: their chunks were loaded as zeros)
      if (ch < 1568) store_planes8(xl + (sy * 18 + xx + 2) * DW3_XPIX + (ch & 7) * 8, DW3_XPLANE, bf3_split4(xv[i]));
    }
    if (tid < 392) {
#pragma unroll
      for (int j = 0; j < 4; ++j) tg[j] = tp[j] > 0.f ? tg[j] : 0.f;
      const Bf3Quad s = bf3_split4(tg);
      const int py = twin / 7, px = twin - py * 7;
      // entry j: [g0 | g1 << 16, g2 | argmax << 16]
      const uint32_t p0[4] = {s.p0.x & 0xffffu, s.p0.x >> 16, s.p0.y & 0xffffu, s.p0.y >> 16};
      const uint32_t p1[4] = {s.p1.x & 0xffffu, s.p1.x >> 16, s.p1.y & 0xffffu, s.p1.y >> 16};
      const uint32_t p2[4] = {s.p2.x & 0xffffu, s.p2.x >> 16, s.p2.y & 0xffffu, s.p2.y >> 16};
      u32x4_x3 e0, e1;
      e0[0] = p0[0] | (p1[0] << 16); e0[1] = p2[0] | (((tav >> 0) & 0xffu) << 16);
      e0[2] = p0[1] | (p1[1] << 16); e0[3] = p2[1] | (((tav >> 8) & 0xffu) << 16);
      e1[0] = p0[2] | (p1[2] << 16); e1[1] = p2[2] | (((tav >> 16) & 0xffu) << 16);
      e1[2] = p0[3] | (p1[3] << 16); e1[3] = p2[3] | (((tav >> 24) & 0xffu) << 16);
      char* dst = pw + ((py * 9 + px) * 32 + tcg * 4) * 8;
      *reinterpret_cast<u32x4_x3*>(dst) = e0;
      *reinterpret_cast<u32x4_x3*>(dst + 16) = e1;
      if (bias_role) { gsum[0] += tg[0]; gsum[1] += tg[1]; gsum[2] += tg[2]; gsum[3] += tg[3]; }
    }
  };

  f32x4 acc[5];
#pragma unroll
  for (int dx = 0; dx < 5; ++dx) acc[dx] = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane part of the transposing reads of X: k-row 8 q + (r >> 2) of the step, columns 4 (r & 3) .. + 3 of the cin tile
  const char* xa = xl + (8 * q + (r >> 2)) * DW3_XPIX + (mt * 16 + 4 * (r & 3)) * 2;
  // the lane's four windows of each of its k-steps (grid positions 32 s + 8 q + 2 j, s = kq + 2 h): LDS offset of the entry
  // of its cout (16 bits each, two per register), and the parity of the grid row (one bit each)
  uint32_t woff[4][2], wpar = 0;
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pos = 32 * (kq + 2 * h) + 8 * q + 2 * j;
      const int y = pos / 18, xx = pos - y * 18;
      const uint32_t o = (((y >> 1) * 9 + (xx >> 1)) * 32 + nt * 16 + r) * 8;
      if (j & 1) woff[h][j >> 1] |= o << 16; else woff[h][j >> 1] = o;
      wpar |= (uint32_t)(y & 1) << (4 * h + j);
    }

  if (nimg > 0) load(b0);
  for (int im = 0; im < nimg; ++im) {
    __syncthreads();  // the previous image has been consumed (first pass: the zero fill is done)
    store();
    __syncthreads();
    if (im + 1 < nimg) load(b0 + im + 1);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int s = kq + 2 * h;  // k-step: grid positions 32 s .. 32 s + 31
      // the B fragment of this wave's cout tile, built from the pooled entries
      uint2 ent[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ent[j] = *reinterpret_cast<const uint2*>(pw + ((woff[h][j >> 1] >> (16 * (j & 1))) & 0xffffu));
      u32x4_x3 f0, f1, f2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t m = (ent[j].y >> 16) - 2u * ((wpar >> (4 * h + j)) & 1u);  // 0: first position, 1: second, else none
        const uint32_t g0 = ent[j].x & 0xffffu, g1 = ent[j].x >> 16, g2 = ent[j].y & 0xffffu;
        f0[j] = m == 0u ? g0 : (m == 1u ? g0 << 16 : 0u);
        f1[j] = m == 0u ? g1 : (m == 1u ? g1 << 16 : 0u);
        f2[j] = m == 0u ? g2 : (m == 1u ? g2 << 16 : 0u);
      }
      const bf16x8 bfr[3] = {__builtin_bit_cast(bf16x8, f0), __builtin_bit_cast(bf16x8, f1), __builtin_bit_cast(bf16x8, f2)};
      // A: per plane ONE 12-position run per lane (three transposing reads); the fragment of tap dx is positions dx .. dx + 7
      // of it -- even dx: registers as they are, odd dx: four funnel shifts (nine reads per plane if every tap read its own)
      bf16x8 a[5];
#define DW3_LDA(P) do {                                                                                             \
        const char* ap = xa + (P) * DW3_XPLANE + 32 * s * DW3_XPIX;                                                  \
        const s16x4_x3 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_x3*)(ap));                      \
        const s16x4_x3 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_x3*)(ap + 4 * DW3_XPIX));       \
        const s16x4_x3 t2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_x3*)(ap + 8 * DW3_XPIX));       \
        const uint2 u0 = __builtin_bit_cast(uint2, t0), u1 = __builtin_bit_cast(uint2, t1), u2 = __builtin_bit_cast(uint2, t2);              \
        const uint32_t L[6] = {u0.x, u0.y, u1.x, u1.y, u2.x, u2.y};                                                  \
        uint32_t S[5];                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) S[i] = __builtin_amdgcn_alignbit(L[i + 1], L[i], 16);          \
        a[0] = __builtin_bit_cast(bf16x8, (u32x4_x3){L[0], L[1], L[2], L[3]});                                       \
        a[1] = __builtin_bit_cast(bf16x8, (u32x4_x3){S[0], S[1], S[2], S[3]});                                       \
        a[2] = __builtin_bit_cast(bf16x8, (u32x4_x3){L[1], L[2], L[3], L[4]});                                       \
        a[3] = __builtin_bit_cast(bf16x8, (u32x4_x3){S[1], S[2], S[3], S[4]});                                       \
        a[4] = __builtin_bit_cast(bf16x8, (u32x4_x3){L[2], L[3], L[4], L[5]});                                       \
      } while (0);
#define DW3_PROD(PB) _Pragma("unroll") for (int dx = 0; dx < 5; ++dx) acc[dx] = mfma16bf(a[dx], bfr[PB], acc[dx]);
      DW3_LDA(2) DW3_PROD(0)
      DW3_LDA(1) DW3_PROD(1) DW3_PROD(0)
      DW3_LDA(0) DW3_PROD(2) DW3_PROD(1) DW3_PROD(0)
#undef DW3_LDA
#undef DW3_PROD
      __builtin_amdgcn_sched_barrier(0);  // (the next step's reads stay behind this step's products: register room)
    }
  }
  // bias gradient: LDS atomics of the task threads, then one 32-lane atomic per workgroup
  if (bias_role && tid < 392) {
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&bsum[tcg * 4 + j], gsum[j]);
  }
  __syncthreads();
  if (bias_role && tid < 32) atomicAdd(&db2[c * 32 + tid], bsum[tid]);
  __syncthreads();
  // combine the two k groups' partial sums (the images are dead), then one atomic per element and workgroup
  float* red = reinterpret_cast<float*>(smem);
  if (kq == 1) {
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) *reinterpret_cast<f32x4*>(red + ((wave * 5 + dx) * 64 + lane) * 4) = acc[dx];
  }
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + (((wave + 1) * 5 + dx) * 64 + lane) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cin = mt * 16 + 4 * q + i, cout = c * 32 + nt * 16 + r;
        atomicAdd(&dW[((long)(dy * 5 + dx) * 32 + cin) * 64 + cout], acc[dx][i] + v[i]);
      }
    }
  }
}

// conv2 dW (workgroups [0, n2)), conv1 dW/db (the next n1) and optionally trailing optimizer workgroups in one launch (the
// roles of conv_dw_duo_f32_kernel, f32_conv2.hip)
struct ConvDwDuoX3 {
  const float* pool1; const float* dpool2; const float* pool2; const uint8_t* amax2; float* dW2; float* db2;
  int B, n2;
  const float* images; const int* perm; const int* step_ptr; int spe;
  const float* dpool1; const uint8_t* amax1; float* dW1; float* db1;
  int n1;
  OptimParams adam;  // adam.n == 0: no optimizer workgroups
  int nadam;
};
constexpr int DUOX3_LDS = DW3_LDS > CONV1_DW_LDS_FLOATS * 4 ? DW3_LDS : CONV1_DW_LDS_FLOATS * 4;
static_assert(DUOX3_LDS <= 81920, "two workgroups per CU");
}  // namespace

__global__ __launch_bounds__(512, 4) void conv_dw_duo_x3_kernel(const ConvDwDuoX3 a) {
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  const int id = blockIdx.x;
  if (id < a.n2)
    conv2_dw_x3_body(smem3, id, a.pool1, a.dpool2, a.pool2, a.amax2, a.dW2, a.db2, a.B);
  else if (id >= a.n2 + a.n1) {
    if (a.adam.wsh_world > 0)  // (launch-uniform) "wshard": this rank's slice only, stored into every rank's next-parity buffer
      adam_tf_body<false, true>(a.adam, id - a.n2 - a.n1, a.nadam, 512);
    else
      adam_tf_body(a.adam, id - a.n2 - a.n1, a.nadam, 512);
  } else
    conv1_dw_body(reinterpret_cast<float*>(smem3), id - a.n2, a.n1, a.images, a.perm, a.step_ptr, a.spe, a.B, a.dpool1,
                  a.amax1, a.dW1, a.db1);
}

__global__ __launch_bounds__(256) void conv2_pack3_kernel(const float* __restrict__ W, char* __restrict__ Wf3, char* __restrict__ Wb3) {
  conv2_pack3_body(W, Wf3, Wb3, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ host side
// Process-wide switch: 1 = the fp32 conv2 forward / data gradient run as bf16x6 (default), 0 = exact-fp32 MFMA.
static int g_f32_emulation = 1;
DTF_API void dtf_f32_emulation_set(int on) { g_f32_emulation = on ? 1 : 0; }
DTF_API int dtf_f32_emulation() { return g_f32_emulation; }

static bool g_x3_attr_set = false;
static char* g_pack3 = nullptr;  // standalone entry points: [Wf3 | Wb3]
static int x3_init() {
  if (g_x3_attr_set) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_fwd_x3_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_bwd_data_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          D3_LDS);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dw_duo_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          DUOX3_LDS);
  if (e != hipSuccess) return (int)e;
  g_x3_attr_set = true;
  return 0;
}

// Both filter gradients of the conv stack (+ the hosted optimizer range) in one launch: the arguments of dtf_conv_dw_duo_f32,
// which validates them and forwards here when the emulation switch is on.
DTF_API int dtf_conv_dw_duo_x3(const float* dpool2, const float* pool2, const void* amax2, const float* pool1, float* dW2,
                               float* db2, int B, const float* images, const int* perm, const int* step_ptr, int spe,
                               const float* dpool1, const void* amax1, float* dW1, float* db1, const OptimParams* adam,
                               hipStream_t st) {
  if (B <= 0 || (perm && spe <= 0)) return ERR_SHAPE;
  const int rc = x3_init();
  if (rc) return rc;
  ConvDwDuoX3 a;
  a.pool1 = pool1; a.dpool2 = dpool2; a.pool2 = pool2; a.amax2 = (const uint8_t*)amax2; a.dW2 = dW2; a.db2 = db2;
  a.B = B; a.n2 = dtf_cdiv(B, DW3_IPW) * 10;
  a.images = images; a.perm = perm; a.step_ptr = step_ptr; a.spe = spe;
  a.dpool1 = dpool1; a.amax1 = (const uint8_t*)amax1; a.dW1 = dW1; a.db1 = db1;
  a.n1 = dW1 ? (B < 128 ? B : 128) : 0;
  a.adam = OptimParams{};
  a.nadam = 0;
  if (adam) {
    a.adam = *adam;
    a.adam.ticket = nullptr;  // (the step is advanced by the optimizer launch that covers the rest of the arena)
    a.nadam = 512 - (a.n2 + a.n1) > 64 ? 512 - (a.n2 + a.n1) : 64;  // (fill the resident slots: two workgroups per CU)
  }
  hipLaunchKernelGGL(conv_dw_duo_x3_kernel, dim3(a.n2 + a.n1 + a.nadam), dim3(512), DUOX3_LDS, st, a);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_conv2_pack3(const float* W, void* Wf3, void* Wb3, hipStream_t st) {
  if (!W || !Wf3 || !Wb3) return ERR_SHAPE;
  hipLaunchKernelGGL(conv2_pack3_kernel, dim3(25), dim3(256), 0, st, W, (char*)Wf3, (char*)Wb3);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_conv2_fwd_x3_packed(const float* x, const void* Wf3, const float* bias, float* out, void* amax, int B,
                                    hipStream_t st) {
  if (B <= 0 || !Wf3) return ERR_SHAPE;
  const int rc = x3_init();
  if (rc) return rc;
  hipLaunchKernelGGL(conv2_fwd_x3_kernel, dim3(dtf_cdiv(49L * B, F3_WIN)), dim3(512), F3_LDS, st, x, (const char*)Wf3, bias, out,
                     (uint8_t*)amax, B);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_conv2_bwd_data_x3_packed(const float* dpool2, const float* pool2, const void* amax2, const void* Wb3,
                                         const float* pool1, float* dpool1, int B, hipStream_t st) {
  if (B <= 0 || !Wb3) return ERR_SHAPE;
  const int rc = x3_init();
  if (rc) return rc;
  hipLaunchKernelGGL(conv2_bwd_data_x3_kernel, dim3(dtf_cdiv(196L * B, D3_POS)), dim3(512), D3_LDS, st, dpool2, pool2,
                     (const uint8_t*)amax2, (const char*)Wb3, pool1, dpool1, B);
  DTF_CHECK_LAUNCH();
  return 0;
}

// standalone forms (tests, evaluation outside the step executor): pack into a process-wide scratch copy first
static int pack3_scratch(const float* W, hipStream_t st) {
  if (!g_pack3 && hipMalloc(&g_pack3, 2 * CONV2_W3_BYTES) != hipSuccess) return -12;
  return dtf_conv2_pack3(W, g_pack3, g_pack3 + CONV2_W3_BYTES, st);
}
DTF_API int dtf_conv2_fwd_x3(const float* x, const float* W, const float* bias, float* out, void* amax, int B, hipStream_t st) {
  const int rc = pack3_scratch(W, st);
  if (rc) return rc;
  return dtf_conv2_fwd_x3_packed(x, g_pack3, bias, out, amax, B, st);
}
DTF_API int dtf_conv2_bwd_data_x3(const float* dpool2, const float* pool2, const void* amax2, const float* W,
                                  const float* pool1, float* dpool1, int B, hipStream_t st) {
  const int rc = pack3_scratch(W, st);
  if (rc) return rc;
  return dtf_conv2_bwd_data_x3_packed(dpool2, pool2, amax2, g_pack3 + CONV2_W3_BYTES, pool1, dpool1, B, st);
}
