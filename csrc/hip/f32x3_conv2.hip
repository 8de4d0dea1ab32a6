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
#include "dtf_api.h"

// -DDTF_CONV_STAMPS (diagnostic builds only, scripts/dbg/x3_conv_stamps.py): thread 0 of every workgroup records (shader clock,
// 100 MHz real-time clock) pairs at phase boundaries into g_x3_stamps[workgroup][16][2]
#ifdef DTF_CONV_STAMPS
__device__ unsigned long long* g_x3_stamps;
DTF_API int dtf_dbg_set_x3_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x3_stamps), &p, sizeof(p)); }
#define X3STAMP(i) do { if (threadIdx.x == 0 && g_x3_stamps) { g_x3_stamps[blockIdx.x * 32 + 2 * (i)] = __builtin_amdgcn_s_memtime(); \
  g_x3_stamps[blockIdx.x * 32 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
// per-wave: loop-end clock at [workgroup][16 + wave], HW_ID at [workgroup][24 + wave]
#define X3STAMP_WAVE() do { if ((threadIdx.x & 63) == 0 && g_x3_stamps) { \
  g_x3_stamps[blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); \
  g_x3_stamps[blockIdx.x * 32 + 24 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); } } while (0)
#else
#define X3STAMP(i) do { } while (0)
#define X3STAMP_WAVE() do { } while (0)
#endif

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
#define X3_PROD(A, Bq, PA, PB)                                                           \
  _Pragma("unroll") for (int mt = 0; mt < 5; ++mt) {                                     \
    acc[mt][0] = mfma16bf(A[mt][PA], Bq[PB][0], acc[mt][0]);                             \
    acc[mt][1] = mfma16bf(A[mt][PA], Bq[PB][1], acc[mt][1]);                             \
  }
// (pinned between scheduling barriers: left to itself the compiler sinks the prefetches issued in front of a k-unit's
// products to a few MFMAs before their use)
#define X3_ALL_PRODUCTS(A, Bq) do {                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    X3_PROD(A, Bq, 2, 0) X3_PROD(A, Bq, 0, 2) X3_PROD(A, Bq, 1, 1) X3_PROD(A, Bq, 1, 0) X3_PROD(A, Bq, 0, 1) X3_PROD(A, Bq, 0, 0) \
    __builtin_amdgcn_sched_barrier(0);                                                   \
  } while (0)
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
  X3STAMP(0);

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
  X3STAMP(1);
  __syncthreads();
  X3STAMP(2);

  // k loop: the fragments of the NEXT k-unit (15 LDS reads, 6 filter loads from L2) are requested before the 60 MFMAs of the
  // current one, in two register sets used in turn.  Phase stamps showed why: with "oldest wave first" issue the older wave
  // of a SIMD runs ahead and the younger one ends up alone for a third of the loop, where a wave that reads, waits and
  // multiplies in turn keeps the pipe 60 % busy; a wave that always has its operands keeps it full on its own.
  auto load_a = [&](bf16x8 (&a)[F3_MT][3], const int t) {
    const int dy = t / 5, dx = t - dy * 5;
    const char* ab = smem3 + dy * F3_ROW + dx * F3_PIX;
    // (plane 2 first: the products that need it come first, and LDS returns in order)
#pragma unroll
    for (int mt = 0; mt < F3_MT; ++mt) a[mt][2] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + 2 * F3_PLANE);
#pragma unroll
    for (int mt = 0; mt < F3_MT; ++mt) a[mt][0] = *reinterpret_cast<const bf16x8*>(ab + abase[mt]);
#pragma unroll
    for (int mt = 0; mt < F3_MT; ++mt) a[mt][1] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + F3_PLANE);
  };
  auto load_b = [&](bf16x8 (&b)[3][2], const int t) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#ifdef DTF_X3_EXP_BHOT  // (timing experiment only, wrong sums: every tap reads tap 0's filter rows -> L1 hits, no L2 stream)
        b[p][nt] = *reinterpret_cast<const bf16x8*>(wb + p * CONV2_W3_PLANE + ((long)(t & 0) * 64 + 16 * nt) * 64);
#else
        b[p][nt] = *reinterpret_cast<const bf16x8*>(wb + p * CONV2_W3_PLANE + ((long)t * 64 + 16 * nt) * 64);
#endif
  };
  bf16x8 acur[F3_MT][3], anxt[F3_MT][3];
  load_a(acur, kq);
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {  // taps kq + 8 j and kq + 8 j + 4
    const int t = kq + 8 * j;
    load_b(bnxt, t + 4);
#ifdef DTF_X3_EXP_NOLDS  // (timing experiment only, wrong sums: no fragment reads inside the loop)
    if (j == 0)
#endif
    load_a(anxt, t + 4);
    X3_ALL_PRODUCTS(acur, bcur);
    const int t2 = t + 8 < 25 ? t + 8 : 24;  // (kq = 0: its seventh tap, 24; the other waves' last prefetch is a harmless read)
    load_b(bcur, t2);
#ifndef DTF_X3_EXP_NOLDS
    load_a(acur, t2);
#endif
    X3_ALL_PRODUCTS(anxt, bnxt);
  }
  if (kq == 0) X3_ALL_PRODUCTS(acur, bcur);  // tap 24 (wave-uniform)

  // ---- combine the four K slices of each channel half through LDS (the image is dead), then bias + ReLU + pool on register
  // quads; tile id = (nh * F3_MT + mt) * 2 + nt, finalised by wave (id % 8)
  X3STAMP_WAVE();
  X3STAMP(3);
  __syncthreads();
  X3STAMP(4);
  float* red = reinterpret_cast<float*>(smem3);
  constexpr int TPW = F3_MT * 2;  // tiles per wave
#pragma unroll
  for (int mt = 0; mt < F3_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<f32x4*>(red + ((wave * TPW + mt * 2 + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();
  X3STAMP(5);
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
  X3STAMP(6);
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
  // Zeros only where the un-pool tasks below do not write (they cover every in-image pixel of the staged rows): the two
  // halo columns on either side of every staged row, and the staged rows outside the image.  Disjoint from the tasks'
  // addresses, so no barrier in between.
  {
    const int nrow = nrow0 + nrow1;
    for (int c = tid; c < nrow * 96; c += 512) {  // (row, plane, halo column 0 1 16 17, 16-byte chunk 0..7)
      const int row = c / 96, rem = c - row * 96, p = rem >> 5, hc = (rem >> 3) & 3;
      *reinterpret_cast<uint4*>(smem3 + p * D3_PLANE + row * D3_ROW + (hc + (hc >> 1) * 14) * D3_PIX + (rem & 7) * 16) =
          make_uint4(0, 0, 0, 0);
    }
    // staged row -> halo row: slot 0 rows [0, nrow0) are halo rows ylo0 .., slot 1 rows are halo rows 0 ..; in-image = 2..15
    for (int row = 0; row < nrow; ++row) {
      const int hy = row < nrow0 ? ylo0 + row : row - nrow0;
      if (hy >= 2 && hy <= 15) continue;  // (workgroup-uniform)
      for (int c = tid; c < 3 * 14 * 8; c += 512) {
        const int p = c / 112, rem = c - p * 112;
        *reinterpret_cast<uint4*>(smem3 + p * D3_PLANE + row * D3_ROW + (2 + (rem >> 3)) * D3_PIX + (rem & 7) * 16) =
            make_uint4(0, 0, 0, 0);
      }
    }
  }
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

  // k loop: two register sets of fragments used in turn, the next unit's requested before the current one's MFMAs (see
  // conv2_fwd_x3_kernel); wave w takes units w, w + 8, ... (7 for waves 0 and 1, 6 for the others)
  auto load_a = [&](bf16x8 (&a)[D3_MT][3], const int u) {
    const int tap = u >> 1;
    const int dy = tap / 5, dx = tap - dy * 5;
    const char* ab = smem3 + dy * D3_ROW + dx * D3_PIX + (u & 1) * 64;
#pragma unroll
    for (int mt = 0; mt < D3_MT; ++mt) a[mt][2] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + 2 * D3_PLANE);
#pragma unroll
    for (int mt = 0; mt < D3_MT; ++mt) a[mt][0] = *reinterpret_cast<const bf16x8*>(ab + abase[mt]);
#pragma unroll
    for (int mt = 0; mt < D3_MT; ++mt) a[mt][1] = *reinterpret_cast<const bf16x8*>(ab + abase[mt] + D3_PLANE);
  };
  auto load_b = [&](bf16x8 (&b)[3][2], const int u) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) b[p][nt] = bfrag(p, nt, u);
  };
  bf16x8 acur[D3_MT][3], anxt[D3_MT][3];
  load_a(acur, wave);
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {  // units wave + 16 j and wave + 16 j + 8
    const int u = wave + 16 * j;
    load_b(bnxt, u + 8);
    load_a(anxt, u + 8);
    X3_ALL_PRODUCTS(acur, bcur);
    const int u2 = u + 16 < 50 ? u + 16 : 49;  // (waves 0 and 1: their seventh unit; the others' last prefetch is a harmless read)
    load_b(bcur, u2);
    load_a(acur, u2);
    X3_ALL_PRODUCTS(anxt, bnxt);
  }
  if (wave < 2) X3_ALL_PRODUCTS(acur, bcur);  // units 48, 49 (wave-uniform)

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

// (The filter gradient stays on the exact-fp32 MFMA, f32_conv2.hip: a bf16x6 form -- pooled dY entries in LDS, transposing
// reads, one 12-position run + funnel shifts for the five shifted A fragments -- was correct but no faster, 29.3 against
// 28.5 us for the role alone: that launch is bound by its per-image staging chain under the hosted optimizer's memory
// traffic, not by the matrix pipe.  docs/DESIGN.md 2d.)

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
  g_x3_attr_set = true;
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
