// "Shard" data parallelism for a wide dense layer at small batch (the MNIST fc1: 12.8 MB of filter gradient from 1.7 MB of
// operands; /root/reference/demo2/train.py:27-29,146-149 puts that traffic on the parameter server every step).
//
// Per-rank work that does not grow with the number of ranks N:
//   1. after the loss head the ranks exchange the OPERANDS of the filter-gradient GEMM, and each rank pulls only what its
//      slice needs: columns [lo, hi) of every peer's pool2 [B,3136] plus the peer's dh [B,1024]  ((1.25/N + 0.4) MB per peer);
//   2. rank r computes rows [lo, hi) of the GLOBAL gradient  dW = sum_p pool2_p[:, lo:hi]^T dh_p  (K = N x batch, M = 3136/N:
//      the same 0.64 GFLOP for every N; f32_fc.hip: K sets of two source ranks, atomics into the slice of G when there is
//      more than one set);
//   3. the slice travels to every rank's landing buffer (12.8/N MB per xGMI link, all links at once) from extra workgroups
//      of the conv2 data-gradient launch: shard_push_body below;
//   4. a one-wave barrier over the ranks (dtf_ipc_sync: "my slice has landed everywhere", bounded wait for the N - 1 peers;
//      waiting inside the 80 KB-LDS workgroups of the next launch instead exhausted the chip's workgroup slots when four
//      ranks time-shared one GPU), then W_fc1's Adam update (hosted by the conv filter-gradient launch as on one GPU) reads
//      the assembled global gradient from the landing buffer.  Every rank applies the same bytes with the same arithmetic, so replicas stay bit-identical and optimizer state stays replicated (checkpoints,
//      check_exchange() and the re-broadcast after a failure work as for every other schedule).
// Ordering: a rank pushes step t+1's slice only after the operand exchange of step t+1, which every peer enters after its
// optimizer launch of step t has finished reading the landing buffer (stream order on the peer) — no "done" handshake.
#pragma once
#include "common.h"

struct ShardPush {
  float* src;          // this rank's slice of the fc1 filter gradient inside G
  long n4;             // float4 units
  float* dst[8];       // the slice's place in every rank's landing buffer (this rank's own included)
  int world, rank;
  int zero_src;        // the slice is accumulated with atomics by several K sets: re-zero it once it has been read
};

// Workgroup `block` of `nblocks` (nthreads each).  Four float4 per lane in flight: 4 loads, then 4 x world stores.
// No flag is raised here: a release from inside the kernel is a write-back of the whole L2 per wave (measured: 2 000 of
// them stretched the hosting launch from 23 to 70 us).  The launch boundary that follows does it once; the one-wave barrier
// kernel after it (dtf_ipc_sync on the landing buffers' communicator) tells the peers and waits for theirs.
__device__ __forceinline__ void shard_push_body(const ShardPush& s, const int block, const int nblocks, const int nthreads) {
  constexpr int U = 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4* __restrict__ src = reinterpret_cast<f32x4*>(s.src);
  for (long base = (long)block * nthreads * U + threadIdx.x; base < s.n4; base += (long)nblocks * nthreads * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * nthreads;
      v[u] = src[i < s.n4 ? i : base];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + (long)u * nthreads;
      if (i >= s.n4) continue;
#pragma unroll
      for (int p = 0; p < 8; ++p)
        if (p < s.world) reinterpret_cast<f32x4*>(s.dst[p])[i] = v[u];
      if (s.zero_src) src[i] = zero;
    }
  }
}

