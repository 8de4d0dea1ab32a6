// Native step executor for the MNIST CNN: the whole train step — batch gather,
// forward, fused loss head, backward, optimizer — is a fixed sequence of 8 kernels issued
// from C++ on one HIP stream and (normally) replayed as a hipGraph, so the ~100 us step is not bound
// by Python/launch overhead.  All per-step state that changes (step counter, batch offset, dropout
// counter, metric slot) lives in device memory, which is what makes the captured graph replayable.
//
// The backward is split in two parts so a data-parallel driver can all-reduce the fc gradient bucket
// (98 % of the bytes) on a communication stream while the conv backward still runs:
//   part 1: conv1 fwd -> conv2 fwd -> fc1 (split-K) -> head -> [fc1 dW | fc2 dW | fc1 bwd-data]   [bucket A ready]
//   part 2: conv2 un-pool + bwd-data (+db2) -> [conv2 dW | conv1 dW/db]                           [bucket B ready]
//   apply : multi-tensor TF-Adam over the flat arena (also zeroes the atomically accumulated grads)
#include "dtf_api.h"
#include "dtf_structs.h"
#include "shard_body.h"

struct MnistEngine {
  int B, maxB, spe, world, double_softmax, fc1_splits, bwd_splits, dw2_splits, head_blocks, ring;
  float lr, beta1, beta2, eps, keep_prob;
  unsigned long long seed;
  const float* images;
  const int* labels;
  const int* perm;
  float *P, *G, *M, *V;
  long n_arena;
  long off_Wfc1, off_bfc1, off_Wfc2, off_bfc2, off_Wc2, off_bc2, off_Wc1, off_bc1;
  int* step;
  unsigned int* ticket;
  float* metrics_ring;
  void *pool1, *amax1, *pool2, *amax2;
  float* slabs;
  void *h_drop, *dh, *dz;
  float* dpool2;
  void* dy2;
  float* dpool1;
  // data-parallel zero-copy exchange (both null otherwise): where the fc1 filter gradient (98 % of the arena, offset 0)
  // is stored instead of G, and where the optimizer reads the reduced gradients of the whole arena from
  float* G_fc1_dst;
  const float* G_src;
  // data-parallel "gather" schedule (gather_n = world size, 0 = off): every rank's pool2 / dh as mapped into this
  // process (exchange buffers of the IpcComm at gather_comm); the fc1 filter gradient is then computed globally
  const void* gather_A[8];
  const void* gather_B[8];
  const void* gather_comm;
  int gather_n, gather_rank;
  void *wc2f, *wc2b;   // packed bf16 copies of conv2's filter (conv2_pack.h), rewritten by the first kernel of every forward
  void* gather_local;  // [gather_n][gather_bytes]: local copies of the peers' operands (what gather_A/B point at for peers)
  long gather_bytes;   // bytes of (pool2, dh) per rank
  // reference precision (demo1/train.py computes in float32): pool1 / pool2 / h_drop / dh are fp32, dz is fp32 [B][16], and
  // every GEMM-shaped op runs on the exact-fp32 MFMA (f32_gemm.h, f32_conv2.hip); 0 = bf16 operands, fp32 accumulate
  int f32;
  // device words [skip flag, skipped-step count] shared with the IPC gradient exchange (OptimParams::skip); may be null
  unsigned* skip;
  // data-parallel "shard" schedule (shard_body.h; shard_n = world size, 0 = off; fp32 only; rides on the gather fields for the
  // operand exchange): this rank computes rows [shard_lo, shard_lo + shard_w) of the GLOBAL fc1 filter gradient and pushes
  // them into every rank's landing buffer (the result buffers of the IpcComm at shard_comm), where W_fc1's Adam update reads
  const void* shard_comm;
  int shard_n, shard_lo, shard_w;
  // "wshard" variant of it (wshard != 0; the shard fields are set as above): WEIGHTS travel instead of gradients, and the
  // optimizer state of W_fc1 is sharded with the gradient.
  //  * W_fc1 lives in the two buffers of the IpcComm at shard_comm (stage[] = parity 0, result[] = parity 1) instead of in P;
  //    the fc1 forward and the fc1 data gradient of step c (c = device step counter) read buffer c & 1;
  //  * rank r computes rows [lo, hi) of the global gradient (as under "shard"), applies Adam to THOSE rows only (M / V of
  //    the other rows are the other ranks' business: 91 MB of optimizer traffic become 91 / N MB) inside the conv
  //    filter-gradient launch, and stores the new rows straight into buffer (c + 1) & 1 of EVERY rank (12.8 / N MB per xGMI
  //    link, all links at once): no push role, no landing barrier;
  //  * the wait of the small all-reduce that ends the step (every rank has published its remaining gradients, stream-ordered
  //    after its filter-gradient launch) is also the proof that every slice of the next buffer has landed.
  // Hazards: buffer (c + 1) & 1 was last read in step c - 1; a peer writes it in step c only after the operand exchange of
  // step c, which this rank enters after its step c - 1 has been enqueued in full.  A skipped step leaves the counter, and
  // with it the buffer every rank reads next, where it was.
  // P[W_fc1], and M / V outside [lo, hi), go stale: dtf_mnist_wshard_store() brings P up to date; the optimizer slices are
  // exchanged by the trainer (parallel/ddp.py: materialize()).
  int wshard;
  // wshard_rows_a > 0: the first wshard_rows_a of this rank's rows are updated (and stored to every rank) from spare
  // workgroups of the conv2 DATA-gradient launch, the rest from the filter-gradient launch as above: the stores of 12.8 / N
  // MB per xGMI link then have two launches (~52 us) to drain instead of one (~29 us), which is what 2 and 4 GPUs need.  On
  // one GPU, or with 8, it only costs (the data-gradient launch is MFMA-bound on every CU); the start-up trial decides.
  int wshard_rows_a;
  // "ipc-gather2" (early_push != 0, plain gather schedule only, fp32): this rank's pool2 is stored into every peer's landing
  // block from spare workgroups of the fc1 forward launch, so it travels while the fc1 forward and the loss head run; the
  // exchange after the head then moves dh only (ipc_allreduce.hip: dtf_ipc_sync_gather_pushed).
  int early_push;
};

// sizeof of the structs that distributed_tensorflow_amd/native.py mirrors with ctypes (checked when the library is loaded:
// a mirror that is shorter than the struct makes every by-value kernel argument read past the caller's buffer)
DTF_API void dtf_struct_sizes(int* out) {
  out[0] = (int)sizeof(HeadParams);
  out[1] = (int)sizeof(OptimParams);
  out[2] = (int)sizeof(MnistEngine);
}

// "shard" schedule: source ranks per K set of the slice GEMM (one set: plain stores; more: atomics into the zeroed slice)
static constexpr int SHARD_RANKS_PER_SET = 2;

#define DTF_TRY(x)            \
  do {                        \
    int _r = (x);             \
    if (_r != 0) return _r;   \
  } while (0)

// fp32 path: conv2 forward / data gradient as bf16x6 (f32x3_conv2.hip) when the switch is on and the engine carries the
// three-plane filter copies (wc2f / wc2b sized 3 x 102 400 B by engine/mnist_engine.py)
static bool x3_convs(const MnistEngine* e) { return e->f32 && dtf_f32_emulation() && e->wc2f && e->wc2b; }

static int forward_to_slabs(const MnistEngine* e, const float* images, const int* perm, const int* step, int n,
                            int splits, bool need_argmax, hipStream_t st, const ShardPush* push = nullptr) {
  if (e->f32) {
    if (x3_convs(e)) {  // bf16x6: conv2's filter planes are rewritten by leading workgroups of the conv1 launch
      DTF_TRY(dtf_conv1_fwd_f32_pack3(images, perm, step, e->spe, n, e->P + e->off_Wc1, e->P + e->off_bc1, (float*)e->pool1,
                                      need_argmax ? e->amax1 : nullptr, e->P + e->off_Wc2, e->wc2f, e->wc2b, st));
      DTF_TRY(dtf_conv2_fwd_x3_packed((const float*)e->pool1, e->wc2f, e->P + e->off_bc2, (float*)e->pool2,
                                      need_argmax ? e->amax2 : nullptr, n, st));
    } else {
      DTF_TRY(dtf_conv1_fwd_f32(images, perm, step, e->spe, n, e->P + e->off_Wc1, e->P + e->off_bc1, (float*)e->pool1,
                                need_argmax ? e->amax1 : nullptr, st));
      DTF_TRY(dtf_conv2_fwd_f32((const float*)e->pool1, e->P + e->off_Wc2, e->P + e->off_bc2, (float*)e->pool2,
                                need_argmax ? e->amax2 : nullptr, n, st));
    }
    if (e->wshard) {
      float* w[16];
      DTF_TRY(dtf_ipc_wshard_bufs(e->shard_comm, 0, 3136L * 1024, w));
      if (e->shard_n < 1 || e->gather_rank < 0 || e->gather_rank >= e->shard_n) return -22;
      return dtf_fc_fwd_f32_sel((const float*)e->pool2, w[e->gather_rank], w[8 + e->gather_rank], e->step, e->slabs, n, 1024,
                                3136, splits, st);
    }
    if (push)
      return dtf_fc_fwd_f32_push((const float*)e->pool2, e->P + e->off_Wfc1, e->slabs, n, 1024, 3136, splits, push, st);
    return dtf_fc_fwd_f32((const float*)e->pool2, e->P + e->off_Wfc1, e->slabs, n, 1024, 3136, splits, nullptr, 0, st);
  }
  if (!e->wc2f || !e->wc2b) return -22;
  DTF_TRY(dtf_conv1_fwd_pack(images, perm, step, e->spe, n, e->P + e->off_Wc1, e->P + e->off_bc1, e->pool1,
                             need_argmax ? e->amax1 : nullptr, e->P + e->off_Wc2, e->wc2f, e->wc2b, st));
  DTF_TRY(dtf_conv2_fwd_lds_packed(e->pool1, e->wc2f, e->P + e->off_bc2, e->pool2, need_argmax ? e->amax2 : nullptr, n, st));
  DTF_TRY(dtf_fc_fwd(e->pool2, 0, e->P + e->off_Wfc1, e->slabs, n, 1024, 3136, splits, nullptr, 0, st));
  return 0;
}

static int effective_splits(int K, int splits, int BK = 64) {  // mirrors base_params() in gemm_api.hip / f32_fc.hip
  const int ksteps = (K + BK - 1) / BK;
  const int kchunk = ((ksteps + splits - 1) / splits) * BK;
  return (K + kchunk - 1) / kchunk;
}

// The backward is a small DAG, not a chain: the fc filter gradients only feed the optimizer, and the two conv filter
// gradients are independent of each other.  Running such branches on forked streams (parallel graph branches) was
// measured and rejected: every cross-queue edge costs 5-10 us on this ~120 us step, as much as the overlap saves.
// Instead independent kernels are fused HORIZONTALLY into one launch on the single stream:
//   head -> [fc1 dW | fc2 dW | fc1 bwd-data] -> conv2 bwd-data -> [conv2 dW | conv1 dW] -> Adam
static bool trio_ok(const MnistEngine* e) {
  if (e->f32) return effective_splits(1024, e->bwd_splits, 32) > 1;  // (the fp32 filter-gradient GEMMs walk K = batch in slabs)
  return e->B <= 128 && effective_splits(1024, e->bwd_splits) > 1;
}

DTF_API int dtf_mnist_fwd_bwd(const MnistEngine* e_in, int parts, hipStream_t st) {
  const int B = e_in->B;
  if (B <= 0 || B > e_in->maxB) return -22;
  // gather schedule: this step's pool2 / dh live in this rank's exchange buffer, where the peers read them in place
  // (evaluation keeps the engine's own, larger buffers)
  MnistEngine loc = *e_in;
  const MnistEngine* e = &loc;
  if (loc.gather_n > 0) {
    if (loc.f32 && B % 20) return -22;  // (fp32: a K slab of the filter-gradient GEMM must not straddle two ranks' operands)
    if (!trio_ok(e_in) || loc.gather_rank < 0 || loc.gather_rank >= loc.gather_n || loc.gather_n > 8) return -22;
    loc.pool2 = const_cast<void*>(loc.gather_A[loc.gather_rank]);
    loc.dh = const_cast<void*>(loc.gather_B[loc.gather_rank]);
  }
  const bool shard = loc.shard_n > 0;
  if (shard && (!loc.f32 || loc.shard_n != loc.gather_n || !loc.shard_comm || loc.shard_lo < 0 || loc.shard_w <= 0 ||
                loc.shard_lo % 4 || loc.shard_w % 4 || loc.shard_lo + loc.shard_w > 3136 || loc.G_fc1_dst || loc.G_src))
    return -22;
  // parts bit 2 (value 4): W_fc1's Adam update rides along with the conv filter-gradient launch at the end of part 2 (its
  // gradient is final after part 1 and W_fc1 has been read by then); the caller finishes with dtf_mnist_apply_part(e, 2).
  // Only where this rank's fc1 gradient is the GLOBAL one: a single replica, or the gather schedule.
  const bool early_adam = (parts & 4) != 0;
  parts &= 3;
  if (shard && (parts & 2) && !early_adam) return -22;  // (the landing buffer is read by the hosted W_fc1 update only)
  const bool wshard = loc.wshard != 0;
  if (wshard && (!shard || e->off_Wfc1 != 0)) return -22;
  float* wsh[16] = {};  // [parity][rank]: W_fc1 in every rank's two buffers
  if (wshard) DTF_TRY(dtf_ipc_wshard_bufs(e->shard_comm, 0, 3136L * 1024, wsh));
  if (early_adam && ((e->world > 1 && e->gather_n == 0) || e->G_fc1_dst || e->G_src || e->off_Wfc1 != 0 || e->off_bfc1 % 4))
    return -22;
  if (e->G_fc1_dst && e->off_Wfc1 != 0) return -22;
  float* const dW_fc1 = e->G_fc1_dst ? e->G_fc1_dst : e->G + e->off_Wfc1;
  const bool early_push = loc.early_push != 0 && loc.gather_n > 0;
  if (early_push && (shard || !loc.f32)) return -22;
  if (parts & 1) {
    ShardPush opush{};
    if (early_push)  // (pool2 of this step is written into this rank's stage buffer by the conv2 forward just before)
      DTF_TRY(dtf_ipc_push_desc(e->gather_comm, (float*)e->pool2, (long)B * 3136, &opush));
    DTF_TRY(forward_to_slabs(e, e->images, e->perm, e->step, B, e->fc1_splits, true, st, early_push ? &opush : nullptr));
    HeadParams hp{};
    hp.slabs = e->slabs;
    hp.nslab = effective_splits(3136, e->fc1_splits, e->f32 ? 32 : 64);
    hp.f32 = e->f32;
    hp.slab_stride = (long)B * 1024;
    hp.b_fc1 = e->P + e->off_bfc1;
    hp.W_fc2 = e->P + e->off_Wfc2;
    hp.b_fc2 = e->P + e->off_bfc2;
    hp.labels_all = e->labels;
    hp.perm = e->perm;
    hp.step_ptr = e->step;
    hp.spe = e->spe;
    hp.B = B;
    hp.keep_prob = e->keep_prob;
    hp.grad_scale = 1.0f / (float)B;
    hp.seed = e->seed;
    hp.double_softmax = e->double_softmax;
    hp.training = 1;
    hp.h_drop = (uint16_t*)e->h_drop;
    hp.dh = (uint16_t*)e->dh;
    hp.dz = (uint16_t*)e->dz;
    // fp32 with the fused fc backward launch: the bias gradients are column sums taken by a few workgroups of that launch
    // (stores, row order) instead of 100 workgroups' atomics on the same 65 cache lines here (2.4 us of the head's 10.3)
    const bool bias_in_trio = e->f32 && (shard || trio_ok(e));
    hp.db_fc2 = bias_in_trio ? nullptr : e->G + e->off_bfc2;
    hp.db_fc1 = bias_in_trio ? nullptr : e->G + e->off_bfc1;
    hp.metrics = e->metrics_ring;
    hp.metrics_ring = e->ring;
    // the split-K fc1 bwd-data accumulates into dpool2 with atomics: the head zeroes it (saves a memset node)
    if (effective_splits(1024, e->bwd_splits, e->f32 ? 32 : 64) > 1) { hp.zero_buf = e->dpool2; hp.zero_n4 = (long)B * 3136 / 4; }
    DTF_TRY(dtf_head(&hp, e->head_blocks, st));
    // fc1 dW[3136][1024] = pool2^T * dh  (stored: 12.8 MB, the one gradient that is never re-zeroed) and
    // fc2 dW[1024][10] = h_drop^T * dz, as one grouped launch when the batch fits a single K slab
    const bool gather = e->gather_n > 0;
    if (gather) {
      // this rank's fc1 operands (pool2 [B,3136], dh [B,1024] in the compute dtype) are complete: tell the peers, wait for
      // theirs, and pull theirs into local memory (one pass over each xGMI link)
      if (e->gather_bytes != (long)B * (3136 + 1024) * (e->f32 ? 4 : 2) || !e->gather_local) return -22;
      if (shard)  // only the columns of pool2 this rank's gradient slice needs, and dh
        DTF_TRY(dtf_ipc_sync_gather_slice(e->gather_comm, e->gather_local, e->gather_bytes, B, 3136 / 4, e->shard_lo / 4,
                                          e->shard_w / 4, (long)B * 784, (long)B * 256, 16, st));
      else if (early_push)
        DTF_TRY(dtf_ipc_sync_gather_pushed(e->gather_comm, e->gather_local, e->gather_bytes, B, 3136 / 4, (long)B * 784,
                                           (long)B * 256, 16, st));
      else
        DTF_TRY(dtf_ipc_sync_gather(e->gather_comm, e->gather_local, e->gather_bytes, 16, st));
    }
    if (e->f32) {
      const float* x2 = (const float*)e->pool2; const float* dh = (const float*)e->dh;
      if (shard) {
        const void* gA[8];
        for (int p = 0; p < 8; ++p)
          gA[p] = p < e->gather_n ? (const void*)((const float*)e->gather_A[p] + e->shard_lo) : nullptr;
        DTF_TRY(dtf_fc_bwd_trio_f32_shard_bias((const float*)e->h_drop, (const float*)e->dz, 16, e->G + e->off_Wfc2, 1024, 10, B, dh,
                                          wshard ? wsh[e->gather_rank] : e->P + e->off_Wfc1, e->dpool2, B, 3136, 1024, e->bwd_splits, gA, e->gather_B,
                                          e->gather_n, SHARD_RANKS_PER_SET, 3136,
                                          e->G + e->off_Wfc1 + (long)e->shard_lo * 1024, e->shard_w, 1024,
                                          wshard ? wsh[8 + e->gather_rank] : nullptr, wshard ? e->step : nullptr,
                                          e->G + e->off_bfc1, e->G + e->off_bfc2, e->metrics_ring, e->ring, e->step, st));
      } else if (trio_ok(e)) {
        DTF_TRY(dtf_fc_bwd_trio_f32_bias(x2, dh, dW_fc1, 3136, 1024, (const float*)e->h_drop, (const float*)e->dz, 16,
                                         e->G + e->off_Wfc2, 1024, 10, B, dh, e->P + e->off_Wfc1, e->dpool2, B, 3136, 1024,
                                         e->bwd_splits, e->gather_A, e->gather_B, gather ? e->gather_n : 0,
                                         e->G + e->off_bfc1, e->G + e->off_bfc2, e->metrics_ring, e->ring, e->step, st));
      } else {
        DTF_TRY(dtf_fc_dw_f32((const float*)e->h_drop, 1024, (const float*)e->dz, 16, e->G + e->off_Wfc2, 1024, 10, B, 1, st));
        DTF_TRY(dtf_fc_dw_f32(x2, 3136, dh, 1024, dW_fc1, 3136, 1024, B, 1, st));
      }
    } else if (trio_ok(e)) {
      // fc1 dW[3136][1024] = pool2^T * dh (stored: 12.8 MB, the one gradient that is never re-zeroed), fc2 dW[1024][10] =
      // h_drop^T * dz and the split-K fc1 data gradient dpool2 += dh * W_fc1^T as one launch
      DTF_TRY(dtf_fc_bwd_trio(e->pool2, e->dh, dW_fc1, 3136, 1024, e->h_drop, e->dz, e->G + e->off_Wfc2, 1024, 10,
                              B, e->dh, e->P + e->off_Wfc1, e->dpool2, B, 3136, 1024, e->bwd_splits, e->gather_A, e->gather_B,
                              gather ? e->gather_n : 0, st));
    } else if (B <= 128) {
      DTF_TRY(dtf_fc_dw_pair(e->pool2, e->dh, dW_fc1, 3136, 1024, e->h_drop, e->dz, e->G + e->off_Wfc2,
                             1024, 10, B, st));
    } else {
      DTF_TRY(dtf_fc_dw(e->h_drop, 0, e->dz, e->G + e->off_Wfc2, 1024, 10, B, 1, st));
      DTF_TRY(dtf_fc_dw(e->pool2, 0, e->dh, dW_fc1, 3136, 1024, B, 1, st));
    }
  }
  if (parts & 2) {
    if (!trio_ok(e)) {  // (otherwise part 1 already produced dpool2)
      if (e->f32) DTF_TRY(dtf_fc_bwd_data_f32((const float*)e->dh, e->P + e->off_Wfc1, e->dpool2, B, 3136, 1024, e->bwd_splits, st));
      else DTF_TRY(dtf_fc_bwd_data(e->dh, e->P + e->off_Wfc1, e->dpool2, B, 3136, 1024, e->bwd_splits, st));
    }
    // un-pool + conv2 bwd-data (+db2) as one LDS-resident kernel, then both conv filter gradients as one launch
    const int ipw = e->dw2_splits > 0 && e->dw2_splits <= 16 ? e->dw2_splits : 4;
    OptimParams fc1{};
    if (early_adam) {
      fc1.P = e->P; fc1.G = e->G; fc1.M = e->M; fc1.V = e->V;
      fc1.n = e->off_bfc1;
      fc1.zero_from = e->off_bfc1;  // (W_fc1's gradient is stored, never accumulated: nothing to re-zero)
      fc1.lr = e->lr; fc1.beta1 = e->beta1; fc1.beta2 = e->beta2; fc1.eps = e->eps;
      fc1.grad_scale = 1.0f / (float)(e->world > 0 ? e->world : 1);
      fc1.step = e->step;
      fc1.skip = e->skip;
    }
    // W_fc1's Adam update rides in the filter-gradient launch.  (Measured alternative for fp32: in the data-gradient
    // launch, whose 245 workgroups leave 267 resident slots — that launch grows from 24.8 to 33.6 us while the
    // filter-gradient launch only shrinks from 34.8 to 29.4 us: the ~20 us of streaming needs the longer host.)
    ShardPush push{};
    if (wshard) {
      // this rank's rows only; read from its buffer of this step's parity, stored into every rank's buffer of the next one
      const long lo = (long)e->shard_lo * 1024;
      fc1.P = nullptr;
      fc1.G = e->G + lo; fc1.M = e->M + lo; fc1.V = e->V + lo;
      fc1.n = (long)e->shard_w * 1024;
      // (several K sets accumulate the slice with atomics: the update re-zeroes it; one set stores it)
      fc1.zero_from = e->shard_n > SHARD_RANKS_PER_SET ? 0 : fc1.n;
      for (int par = 0; par < 2; ++par)
        for (int r = 0; r < e->shard_n; ++r) fc1.Wsh[par][r] = wsh[par * 8 + r] + lo;
      fc1.wsh_world = e->shard_n;
      fc1.wsh_rank = e->gather_rank;
    }
    OptimParams fc1a{};  // "wshard": the share of the rows that rides in the data-gradient launch
    const int rows_a = wshard ? e->wshard_rows_a : 0;
    if (rows_a < 0 || rows_a % 4 || (wshard && rows_a >= e->shard_w)) return -22;
    if (rows_a > 0) {
      const long na = (long)rows_a * 1024;
      fc1a = fc1;
      fc1a.n = na;
      fc1a.zero_from = fc1.zero_from == 0 ? 0 : na;
      fc1.G += na; fc1.M += na; fc1.V += na;
      fc1.n -= na;
      fc1.zero_from = fc1.zero_from == 0 ? 0 : fc1.n;
      for (int par = 0; par < 2; ++par)
        for (int r = 0; r < e->shard_n; ++r) fc1.Wsh[par][r] += na;
    }
    if (shard && !wshard) {
      DTF_TRY(dtf_ipc_shard_desc(e->shard_comm, (long)e->shard_lo * 1024, (long)e->shard_w * 1024,
                                 e->G + e->off_Wfc1 + (long)e->shard_lo * 1024, e->shard_n > SHARD_RANKS_PER_SET, &push));
      fc1.Gsrc = dtf_ipc_landing(e->shard_comm);  // the assembled global gradient (the push below zeroes the slice in G)
    }
    if (x3_convs(e) && rows_a <= 0 && !(shard && !wshard))  // (no hosted role: the planes written by this step's conv1 launch)
      DTF_TRY(dtf_conv2_bwd_data_x3_packed(e->dpool2, (const float*)e->pool2, e->amax2, e->wc2b, (const float*)e->pool1,
                                           e->dpool1, B, st));
    else if (e->f32)
      DTF_TRY(dtf_conv2_bwd_data_f32_push(e->dpool2, (const float*)e->pool2, e->amax2, e->P + e->off_Wc2,
                                          (const float*)e->pool1, e->dpool1, B, rows_a > 0 ? &fc1a : nullptr,
                                          shard && !wshard ? &push : nullptr, st));
    else
      DTF_TRY(dtf_conv2_bwd_data_lds_packed(e->dpool2, e->pool2, e->amax2, e->wc2b, e->pool1, e->dpool1, e->G + e->off_bc2, B,
                                            st));
    // shard schedule: every rank's slice has left with the launch above; one-wave barrier "mine has landed / all have landed"
    if (shard && !wshard) DTF_TRY(dtf_ipc_sync(e->shard_comm, st));
    if (e->f32)
      DTF_TRY(dtf_conv_dw_duo_f32(e->dpool2, (const float*)e->pool2, e->amax2, (const float*)e->pool1, e->G + e->off_Wc2,
                                  e->G + e->off_bc2, B,
                                  e->images, e->perm, e->step, e->spe, e->dpool1, e->amax1, e->G + e->off_Wc1,
                                  e->G + e->off_bc1, early_adam ? &fc1 : nullptr, st));
    else
      DTF_TRY(dtf_conv_dw_duo(e->dpool2, e->pool2, e->amax2, e->pool1, e->G + e->off_Wc2, B, ipw, e->images, e->perm, e->step,
                              e->spe, e->dpool1, e->amax1, e->G + e->off_Wc1, e->G + e->off_bc1, early_adam ? &fc1 : nullptr,
                              st));
  }
  return 0;
}

// which: 0 = whole arena; 2 = everything but W_fc1 (updated by the conv filter-gradient launch, see above) + step counter
static OptimParams optim_range(const MnistEngine* e, int which) {
  const long lo = which == 2 ? e->off_bfc1 : 0;
  OptimParams op{};
  op.P = e->P + lo; op.G = e->G + lo; op.M = e->M + lo; op.V = e->V + lo;
  op.n = e->n_arena - lo;
  op.zero_from = which == 2 ? 0 : e->off_bfc1;  // everything after W_fc1 is accumulated with atomics
  op.lr = e->lr; op.beta1 = e->beta1; op.beta2 = e->beta2; op.eps = e->eps;
  op.grad_scale = 1.0f / (float)(e->world > 0 ? e->world : 1);
  op.step = e->step;
  op.ticket = e->ticket;
  op.metrics_ring = e->metrics_ring;
  op.ring = e->ring;
  op.Gsrc = e->G_src ? e->G_src + lo : nullptr;
  op.skip = e->skip;
  return op;
}
static int apply_range(const MnistEngine* e, int which, hipStream_t st) {
  const OptimParams op = optim_range(e, which);
  return dtf_adam_tf(&op, st);
}

DTF_API int dtf_mnist_apply(const MnistEngine* e, hipStream_t st) { return apply_range(e, 0, st); }
DTF_API int dtf_mnist_apply_part(const MnistEngine* e, int which, hipStream_t st) {
  if (which != 0 && which != 2) return -22;
  if (which == 2 && (e->off_Wfc1 != 0 || e->off_bfc1 % 4 || (e->n_arena - e->off_bfc1) % 4)) return -22;
  return apply_range(e, which, st);
}

// dtf_mnist_apply_part with the gradients of the range summed over the ranks inside the optimizer launch: every rank has
// published G[range] with dtf_ipc_publish on `comm` earlier in the stream (gather schedule: the range after W_fc1).
DTF_API int dtf_mnist_apply_part_ipc(const MnistEngine* e, int which, const void* comm, hipStream_t st) {
  if ((which != 0 && which != 2) || !comm || e->G_src) return -22;
  if (which == 2 && (e->off_Wfc1 != 0 || e->off_bfc1 % 4 || (e->n_arena - e->off_bfc1) % 4)) return -22;
  const OptimParams op = optim_range(e, which);
  return dtf_ipc_adam_oneshot(comm, &op, st);
}

// "wshard" schedule: P[W_fc1] -> both parity buffers of this rank (entering the schedule, or after the arena was rewritten:
// every replica holds the same P then), and the buffer of the current parity -> P[W_fc1] (before anything reads P).
namespace {
__global__ __launch_bounds__(256) void wshard_store_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                           const int* __restrict__ step, float* __restrict__ dst, long n4) {
  const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>((*step & 1) ? w1 : w0);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
    reinterpret_cast<f32x4*>(dst)[i] = src[i];
}
}  // namespace
DTF_API int dtf_mnist_wshard_load(const MnistEngine* e, hipStream_t st) {
  if (!e->wshard || !e->shard_comm || e->off_Wfc1 != 0 || e->gather_rank < 0 || e->gather_rank >= 8) return -22;
  float* w[16];
  DTF_TRY(dtf_ipc_wshard_bufs(e->shard_comm, 0, 3136L * 1024, w));
  for (int par = 0; par < 2; ++par)
    if (hipMemcpyAsync(w[par * 8 + e->gather_rank], e->P + e->off_Wfc1, 3136L * 1024 * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return -1;
  return 0;
}
DTF_API int dtf_mnist_wshard_store(const MnistEngine* e, hipStream_t st) {
  if (!e->wshard || !e->shard_comm || e->off_Wfc1 != 0 || e->gather_rank < 0 || e->gather_rank >= 8) return -22;
  float* w[16];
  DTF_TRY(dtf_ipc_wshard_bufs(e->shard_comm, 0, 3136L * 1024, w));
  hipLaunchKernelGGL(wshard_store_kernel, dim3(1024), dim3(256), 0, st, w[e->gather_rank], w[8 + e->gather_rank], e->step,
                     e->P + e->off_Wfc1, 3136L * 1024 / 4);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Inference / evaluation over n samples (contiguous or via `index`), chunked by the workspace size.
// metrics[2] (loss sum, correct count) must be zeroed by the caller; probs / pred are optional.
DTF_API int dtf_mnist_eval(const MnistEngine* e, const float* images, const int* labels, int n, float* metrics,
                           float* probs, int* pred, hipStream_t st) {
  if (n <= 0) return -22;
  for (int off = 0; off < n; off += e->maxB) {
    const int c = (n - off) < e->maxB ? (n - off) : e->maxB;
    DTF_TRY(forward_to_slabs(e, images + (long)off * 784, nullptr, nullptr, c, 1, false, st));
    HeadParams hp{};
    hp.slabs = e->slabs;
    hp.nslab = 1;
    hp.slab_stride = (long)c * 1024;
    hp.b_fc1 = e->P + e->off_bfc1;
    hp.W_fc2 = e->P + e->off_Wfc2;
    hp.b_fc2 = e->P + e->off_bfc2;
    hp.labels_all = labels + off;
    hp.B = c;
    hp.spe = 1;
    hp.keep_prob = 1.0f;
    hp.grad_scale = 0.f;
    hp.double_softmax = e->double_softmax;
    hp.training = 0;
    hp.f32 = e->f32;
    hp.metrics = metrics;
    hp.metrics_ring = 0;
    hp.probs = probs ? probs + (long)off * 10 : nullptr;
    hp.pred = pred ? pred + off : nullptr;
    const int blocks = c < 1024 ? c : 1024;
    DTF_TRY(dtf_head(&hp, blocks, st));
  }
  return 0;
}

// ------------------------------------------------------------------ hipGraph helpers
// Capture whatever is enqueued on `st` between begin/end (our launchers, torch ops, RCCL calls)
// and replay it with one host call.  Relaxed mode: other threads (checkpoint saver) may keep
// issuing HIP calls while the capture is open.
struct DtfGraph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

DTF_API int dtf_graph_begin(hipStream_t st) { return (int)hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed); }

DTF_API int dtf_graph_end(hipStream_t st, void** out) {
  DtfGraph* g = new DtfGraph{};
  hipError_t e = hipStreamEndCapture(st, &g->graph);
  if (e != hipSuccess) { delete g; return (int)e; }
  e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { hipGraphDestroy(g->graph); delete g; return (int)e; }
  *out = g;
  return 0;
}

DTF_API int dtf_graph_launch(void* h, hipStream_t st) { return (int)hipGraphLaunch(((DtfGraph*)h)->exec, st); }

DTF_API int dtf_graph_num_nodes(void* h) {
  size_t n = 0;
  if (hipGraphGetNodes(((DtfGraph*)h)->graph, nullptr, &n) != hipSuccess) return -1;
  return (int)n;
}

DTF_API void dtf_graph_free(void* h) {
  DtfGraph* g = (DtfGraph*)h;
  if (!g) return;
  hipGraphExecDestroy(g->exec);
  hipGraphDestroy(g->graph);
  delete g;
}

DTF_API int dtf_version() { return 1; }
