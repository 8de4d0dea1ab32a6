// Direct two-shot all-reduce over IPC-mapped peer memory (SURVEY §5.8 "idiomatic alternative" to the RCCL ring):
// xGMI is a full mesh of point-to-point links, so instead of pushing 2(N-1)/N*S bytes around a ring through ONE link
// per GPU, every GPU reduces its own 1/N slice by reading that slice from all peers at once (N-1 links busy
// concurrently, S/N bytes each) and writes the reduced slice back into every peer's result buffer.  Per-link traffic
// is S/N per phase: 13.1 MB of MNIST-CNN gradients on 8 GPUs = 1.6 MB per link per phase.
//
// Memory: each rank owns ONE fine-grained (uncached, hipDeviceMallocFinegrained) allocation shared through
// hipIpcGetMemHandle:  [ ready[8] | done[8] | pad | stage[n_max] | result[n_max] ].  Fine-grained pages are never held
// in a non-coherent L2, so data written by a peer is what the local Adam kernel reads; plain (cached) gradient
// memory could serve stale lines.  Also the only way to run N ranks against the single GPU of a test box (RCCL
// refuses duplicate devices).
//
// One all-reduce = three stream-ordered kernels (capturable in a hipGraph: the epoch lives in device memory):
//   K1 publish : stage_own <- src (the local gradient arena); bumps the epoch.
//   K2 reduce  : tell every peer "my stage holds epoch e"; wait until all peers said so; for my slice: sum the N
//                stages in rank order (bit-identical on every replica by construction: each element is reduced by
//                exactly one rank) and store the sum into every rank's result buffer.
//   K3 collect : tell every peer "my slice of epoch e is written"; wait for all peers; dst <- result_own.
// Hazards: a peer overwrites result_own for epoch e+1 only after I signalled ready(e+1), which is stream-ordered
// after my K3(e) copy; I overwrite stage_own for e+1 only after every peer signalled done(e), i.e. finished reading.
// Every wait is bounded (wall-clock timeout): on expiry the kernel records an error and proceeds, so the grid always
// drains; the host checks the error word.
//
// The same buffers and flags also serve: a zero-copy all-reduce (producer writes stage[], consumer reads result[]), a
// one-shot all-reduce for small buffers, a one-wave stream-ordered barrier, the operand exchange of the "gather"
// data-parallel schedule (barrier + pull), and plain all-gather / broadcast — see the sections below.
#include <hip/hip_runtime.h>
#include <cstring>
#include "common.h"
#include "dtf_structs.h"
#include "adam_body.h"
#include "shard_body.h"
#include "dtf_api.h"

namespace {
constexpr int MAX_RANKS = 8;
constexpr int HDR_BYTES = 256;  // ready[8], done[8], padding

struct IpcComm {
  int rank, world;
  long n_max;                  // capacity in floats (multiple of 4)
  unsigned char* base[MAX_RANKS];  // every rank's shared allocation as mapped into this process
  unsigned* epoch;             // device word (ordinary memory): epoch of the all-reduce in flight
  unsigned* error;             // device word: number of expired waits (== epoch + 1; epoch + 2 is an arrival ticket,
                               // epoch + 3 the error total of all ranks this rank has acted on, epoch + 4 a decision tag)
  unsigned long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
  unsigned* skip;              // optional device words shared with the optimizer (OptimParams::skip): [0] != 0 = "a wait
                               // of this step's exchange expired somewhere: do not apply this step", [1] = steps skipped
};

__device__ __forceinline__ unsigned* ready_flags(unsigned char* b) { return reinterpret_cast<unsigned*>(b); }
__device__ __forceinline__ unsigned* done_flags(unsigned char* b) { return reinterpret_cast<unsigned*>(b + 64); }
__device__ __forceinline__ unsigned* err_pub(unsigned char* b) { return reinterpret_cast<unsigned*>(b + 128); }
__device__ __forceinline__ float* stage_of(unsigned char* b) { return reinterpret_cast<float*>(b + HDR_BYTES); }
__device__ __forceinline__ float* result_of(unsigned char* b, long n_max) { return reinterpret_cast<float*>(b + HDR_BYTES) + n_max; }

// spin until *flag >= want (system-scope acquire) or the timeout expires; returns false on expiry
// (relaxed polling; the caller issues ONE system-scope acquire fence after all its flags have arrived — an acquire
// per poll, or a release per workgroup, costs a cache invalidate / write-back each: measured ~0.09 us per workgroup
// and kernel, which made larger grids slower instead of faster)
__device__ __forceinline__ bool wait_flag(const unsigned* flag, unsigned want, unsigned long long timeout) {
  const unsigned long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
    if (wall_clock64() - t0 > timeout) return false;
    __builtin_amdgcn_s_sleep(8);
  }
  return true;
}

// ---- expired waits: never apply a partial sum.
// A wait that expires lets the kernel proceed (the grid must drain), so the data it guards may be incomplete.  Every rank
// keeps a PUBLIC count of its expired waits in its shared header; at the end of an exchange one thread per rank adds up
// all ranks' counts and, if the total moved since it last looked, raises the skip word the optimizer launch honours:
// the step is not applied (parameters, moments and step counter stay; accumulated gradients are discarded).  A rank
// publishes its count at the end of the phase in which the wait expired, i.e. before it signals the next phase, so a
// peer that gets through the later phases on that rank's (incomplete) data sees the count.  The one hole — a wait that
// expires in the LAST phase is known to that rank only — leaves the others with complete sums: nobody applies a partial
// one, the replicas may differ by one whole update, and DataParallelTrainer.ensure_healthy() re-synchronises them.
// Call from every thread of the block after the phase's waits; `recorder`: this block accounts for the rank.
// (`recorder` is workgroup-uniform.  The peers' counts are read by one lane per peer, all at once: one lane reading them in a
// loop is world - 1 dependent round trips over xGMI, ~10 us per exchange on 8 GPUs.)
__device__ __forceinline__ void settle_waits(const IpcComm& c, const bool timed_out, const bool recorder, const bool decide) {
  __shared__ unsigned nerr, peers, own;
  if (threadIdx.x == 0) { nerr = 0u; peers = 0u; }
  __syncthreads();
  if (timed_out) atomicAdd(&nerr, 1u);
  __syncthreads();
  if (!recorder) return;
  if (threadIdx.x == 0) {
    unsigned mine = __hip_atomic_load(c.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nerr) {
      mine += nerr;
      __hip_atomic_store(c.error, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(err_pub(c.base[c.rank]), mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    own = mine;
  }
  if (!(decide && c.skip)) return;
  if ((int)threadIdx.x < c.world && (int)threadIdx.x != c.rank)
    atomicAdd(&peers, __hip_atomic_load(err_pub(c.base[threadIdx.x]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned sum = own + peers;
    unsigned* seen = c.epoch + 3;
    if (sum != *seen) {
      *seen = sum;
      __hip_atomic_store(c.skip, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Grid-strided copy with COPY_U independent 16-byte loads in flight per lane: at <= 128 workgroups a one-load-per-
// iteration loop keeps only ~0.5 MB in flight, a fraction of what HBM latency x bandwidth needs.
constexpr int COPY_U = 8;
__device__ __forceinline__ void copy_f4(f32x4* __restrict__ dst, const f32x4* __restrict__ src, const long n4) {
  for (long base = (long)blockIdx.x * 256 * COPY_U + threadIdx.x; base < n4; base += (long)gridDim.x * 256 * COPY_U) {
    f32x4 v[COPY_U];
#pragma unroll
    for (int u = 0; u < COPY_U; ++u) {
      const long i = base + u * 256;
      v[u] = src[i < n4 ? i : base];
    }
#pragma unroll
    for (int u = 0; u < COPY_U; ++u) {
      const long i = base + u * 256;
      if (i < n4) dst[i] = v[u];
    }
  }
}

__global__ __launch_bounds__(256) void ipc_publish_kernel(const IpcComm c, const float* __restrict__ src, long n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *c.epoch += 1u;  // read by K2/K3, which are stream-ordered after this kernel
  copy_f4(reinterpret_cast<f32x4*>(stage_of(c.base[c.rank])), reinterpret_cast<const f32x4*>(src), n >> 2);
}

__global__ __launch_bounds__(256) void ipc_reduce_kernel(const IpcComm c, long n) {
  const unsigned e = *c.epoch;
  bool expired = false;
  if (threadIdx.x < c.world) {
    const int p = threadIdx.x;
    // workgroup 0 signals (the publish kernel has completed: stream order); no workgroup depends on another one of
    // this grid, so the grid needs no co-residency
    if (blockIdx.x == 0)
      __hip_atomic_store(&ready_flags(c.base[p])[c.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    expired = !wait_flag(&ready_flags(c.base[c.rank])[p], e, c.timeout_ticks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the peers' stage buffers are read below
  }
  settle_waits(c, expired, blockIdx.x == 0, false);
  const long n4 = n >> 2;
  const long chunk = (n4 + c.world - 1) / c.world;
  const long lo = (long)c.rank * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  // two elements per lane and iteration, all peer loads of both in flight together (2 x world 16-byte requests per lane)
  for (long i0 = lo + (long)blockIdx.x * 512 + threadIdx.x; i0 < hi; i0 += (long)gridDim.x * 512) {
    const long i1 = i0 + 256;
    const bool two = i1 < hi;
    f32x4 v[2][MAX_RANKS];
#pragma unroll
    for (int p = 0; p < MAX_RANKS; ++p)
      if (p < c.world) {
        const f32x4* sp = reinterpret_cast<const f32x4*>(stage_of(c.base[p]));
        v[0][p] = sp[i0];
        v[1][p] = sp[two ? i1 : i0];
      }
    f32x4 s0 = v[0][0], s1 = v[1][0];
#pragma unroll
    for (int p = 1; p < MAX_RANKS; ++p)
      if (p < c.world) { s0 += v[0][p]; s1 += v[1][p]; }
#pragma unroll
    for (int p = 0; p < MAX_RANKS; ++p)
      if (p < c.world) {
        f32x4* rp = reinterpret_cast<f32x4*>(result_of(c.base[p], c.n_max));
        rp[i0] = s0;
        if (two) rp[i1] = s1;
      }
  }
}

__global__ __launch_bounds__(256) void ipc_collect_kernel(const IpcComm c, float* __restrict__ dst, long n) {
  const unsigned e = *c.epoch;
  bool expired = false;
  if (threadIdx.x < c.world) {
    const int p = threadIdx.x;
    // K2 has completed (stream order), so this rank's slice is written everywhere: publish that, then wait for all
    if (blockIdx.x == 0)
      __hip_atomic_store(&done_flags(c.base[p])[c.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    expired = !wait_flag(&done_flags(c.base[c.rank])[p], e, c.timeout_ticks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  settle_waits(c, expired, blockIdx.x == 0, true);
  copy_f4(reinterpret_cast<f32x4*>(dst), reinterpret_cast<const f32x4*>(result_of(c.base[c.rank], c.n_max)), n >> 2);
}

// ---- zero-copy variant: the producers wrote stage[0, rest_off) themselves (the step executor points the big fc1
// filter-gradient GEMM straight at the stage buffer) and the consumer reads result[] in place (the optimizer takes it as
// its gradient source), so both 13 MB local copies disappear; only the small tail [rest_off, rest_off + rest_n) is copied.
__global__ __launch_bounds__(256) void ipc_publish_rest_kernel(const IpcComm c, const float* __restrict__ src_rest,
                                                               long rest_off, long rest_n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *c.epoch += 1u;
  copy_f4(reinterpret_cast<f32x4*>(stage_of(c.base[c.rank]) + rest_off), reinterpret_cast<const f32x4*>(src_rest), rest_n >> 2);
}

// one wave: "my slice is written everywhere" (the reduce kernel has completed: stream order), then wait until every
// peer says the same; the kernels that follow on the stream may read result[] and overwrite stage[]
__global__ __launch_bounds__(64) void ipc_done_kernel(const IpcComm c) {
  const unsigned e = *c.epoch;
  bool expired = false;
  if (threadIdx.x < c.world) {
    const int p = threadIdx.x;
    __hip_atomic_store(&done_flags(c.base[p])[c.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    expired = !wait_flag(&done_flags(c.base[c.rank])[p], e, c.timeout_ticks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  settle_waits(c, expired, true, true);
}

// ---- activation exchange ("gather" schedule).  For a wide dense layer at small batch the operands of the filter-
// gradient GEMM (x^T [K x batch] and dy [batch x N], bf16) are far smaller than the gradient itself (0.8 MB against
// 12.8 MB for the MNIST fc1), and xGMI all-reduce time is per-link bound: so every rank publishes its two operand
// tensors, and every rank computes the GLOBAL gradient  sum_r x_r^T dy_r  itself, reading the peers' operands in
// place over all links at once (gemm_api.hip: fc_bwd_trio_kernel).  Same rank order everywhere -> bit-identical
// replicas.  The operands are produced straight into stage[] (the step executor points its pool2 / dh buffers there); a
// one-wave kernel then bumps the epoch, tells every peer "stage holds epoch e" and waits for all of them.  There is no
// "done" handshake here: the caller runs an all-reduce of the remaining gradients on another IpcComm later in the same
// step, and no rank enters that before it has finished reading — which is what makes overwriting stage[] in the next
// step safe.
__global__ __launch_bounds__(64) void ipc_sync_kernel(const IpcComm c) {
  const unsigned e = *c.epoch + 1u;  // (one wave: every lane reads before lane 0 writes)
  if (threadIdx.x == 0) *c.epoch = e;
  bool expired = false;
  if (threadIdx.x < c.world) {
    const int p = threadIdx.x;
    __hip_atomic_store(&ready_flags(c.base[p])[c.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    expired = !wait_flag(&ready_flags(c.base[c.rank])[p], e, c.timeout_ticks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  settle_waits(c, expired, true, true);
}

// After the barrier: pull every peer's operands (stage[0, n4) in 16-byte units) into local memory, dst[p][n4], once.  The
// GEMM that consumes them re-reads every operand tile from 16-25 workgroups; peer memory is not cached in the reader's
// L2, so reading it in place would put that factor on the xGMI links (15 MB instead of 0.8 MB per peer and step).
__global__ __launch_bounds__(256) void ipc_gather_copy_kernel(const IpcComm c, f32x4* __restrict__ dst, long n4,
                                                             int blocks_per_peer) {
  const int slot = blockIdx.x / blocks_per_peer, sub = blockIdx.x % blocks_per_peer;
  const int p = slot + (slot >= c.rank ? 1 : 0);  // peers only: this rank's own operands are read from its stage buffer
  if (p >= c.world) return;
  const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(stage_of(c.base[p]));
  f32x4* __restrict__ d = dst + (long)p * n4;
  for (long base = (long)sub * 256 * COPY_U + threadIdx.x; base < n4; base += (long)blocks_per_peer * 256 * COPY_U) {
    f32x4 v[COPY_U];
#pragma unroll
    for (int u = 0; u < COPY_U; ++u) {
      const long i = base + u * 256;
      v[u] = src[i < n4 ? i : base];
    }
#pragma unroll
    for (int u = 0; u < COPY_U; ++u) {
      const long i = base + u * 256;
      if (i < n4) d[i] = v[u];
    }
  }
}

// The two kernels above as ONE launch (a launch boundary costs ~2.5 us of a ~110 us step): workgroup 0 announces this
// rank's operands to every peer; each workgroup then waits for ITS peer's announcement only and copies its share of that
// peer's operands.  The epoch word is read by every workgroup before the last one to finish advances it (arrival
// ticket in the device word after `error`), so no workgroup can see the new value early.  Same ordering contract as
// dtf_ipc_sync + dtf_ipc_gather_copy: when the launch has completed, every peer has announced and been read.
// SLICE ("shard" schedule, shard_body.h): of a peer's stage buffer only `rows` row pieces of w4 float4 (at lo4 inside
// rows of pitch4 float4: the columns of pool2 this rank's gradient slice needs) and a contiguous tail (dh) are pulled, to the
// same offsets of the local copy; n4 is unused.
// land_stride4 > 0 ("early push", dtf_ipc_push_desc): the row pieces are NOT pulled from the peer: the peer has pushed its rows
// into this rank's own result buffer (block p of land_stride4 float4, same row layout) during an earlier launch; they are
// copied from there (local traffic) and only the tail crosses the link here.
struct GatherSlice { int rows, pitch4, lo4, w4; long tail_off4, tail_n4; long land_stride4; };
template <bool SLICE>
__global__ __launch_bounds__(256) void ipc_sync_gather_kernel(const IpcComm c, f32x4* __restrict__ dst, long n4,
                                                             int blocks_per_peer, const GatherSlice gs) {
  const unsigned e = *c.epoch + 1u;
  unsigned* ticket = c.epoch + 2;
  if (blockIdx.x == 0 && threadIdx.x < c.world)
    __hip_atomic_store(&ready_flags(c.base[threadIdx.x])[c.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const int slot = blockIdx.x / blocks_per_peer, sub = blockIdx.x % blocks_per_peer;
  const int p = slot + (slot >= c.rank ? 1 : 0);  // peers only
  if (p < c.world) {
    if (threadIdx.x == 0) {
      if (!wait_flag(&ready_flags(c.base[c.rank])[p], e, c.timeout_ticks) && sub == 0) atomicAdd(c.error, 1u);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the peer's stage buffer is read below
    }
    __syncthreads();
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(stage_of(c.base[p]));
    const f32x4* __restrict__ land = SLICE && gs.land_stride4 > 0
        ? reinterpret_cast<const f32x4*>(result_of(c.base[c.rank], c.n_max)) + (long)p * gs.land_stride4 : nullptr;
    const long body_units = SLICE ? (long)gs.rows * gs.w4 : 0;
    const long total = SLICE ? (long)gs.rows * gs.w4 + gs.tail_n4 : n4;
    f32x4* __restrict__ d = dst + (long)p * (SLICE ? gs.tail_off4 + gs.tail_n4 : n4);
    // unit i of the pull -> float4 offset inside the (identically laid out) source and destination buffers
    auto at = [&](long i) -> long {
      if (!SLICE) return i;
      const long body = (long)gs.rows * gs.w4;
      return i < body ? (i / gs.w4) * gs.pitch4 + gs.lo4 + i % gs.w4 : gs.tail_off4 + (i - body);
    };
    for (long base = (long)sub * 256 * COPY_U + threadIdx.x; base < total; base += (long)blocks_per_peer * 256 * COPY_U) {
      f32x4 v[COPY_U];
#pragma unroll
      for (int u = 0; u < COPY_U; ++u) {
        const long i = base + u * 256;
        const long j = i < total ? i : base;
        v[u] = (land && j < body_units) ? land[at(j)] : src[at(j)];
      }
#pragma unroll
      for (int u = 0; u < COPY_U; ++u) {
        const long i = base + u * 256;
        if (i < total) d[at(i)] = v[u];
      }
    }
  }
  __shared__ unsigned last, peers;
  if (threadIdx.x == 0) {
    last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    peers = 0u;
  }
  __syncthreads();
  if (!last) return;  // (workgroup-uniform)
  // every workgroup's waits are over: publish this rank's error count, look at the peers' (one lane per peer, see settle_waits)
  unsigned mine = 0u;
  if (threadIdx.x == 0) {
    *ticket = 0u;
    *c.epoch = e;
    mine = __hip_atomic_load(c.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(err_pub(c.base[c.rank]), mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (!c.skip) return;
  if ((int)threadIdx.x < c.world && (int)threadIdx.x != c.rank)
    atomicAdd(&peers, __hip_atomic_load(err_pub(c.base[threadIdx.x]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned sum = mine + peers;
    unsigned* seen = c.epoch + 3;
    if (sum != *seen) {
      *seen = sum;
      __hip_atomic_store(c.skip, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// dst <- the first n4 16-byte units of rank `peer`'s stage buffer (general-purpose all-gather / broadcast building block)
__global__ __launch_bounds__(256) void ipc_copy_from_kernel(const IpcComm c, int peer, f32x4* __restrict__ dst, long n4) {
  copy_f4(dst, reinterpret_cast<const f32x4*>(stage_of(c.base[peer])), n4);
}

// One-shot all-reduce for small buffers: after the publish kernel every rank sums ALL ranks' stage buffers itself (rank
// order: bit-identical everywhere) straight into dst — one exchange phase and one kernel less than the two-shot form,
// at N x the read volume.  No "done" handshake: see dtf_ipc_allreduce_oneshot for the contract.
__global__ __launch_bounds__(256) void ipc_reduce_all_kernel(const IpcComm c, float* __restrict__ dst, long n) {
  const unsigned e = *c.epoch;
  bool expired = false;
  if (threadIdx.x < c.world) {
    const int p = threadIdx.x;
    if (blockIdx.x == 0)
      __hip_atomic_store(&ready_flags(c.base[p])[c.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    expired = !wait_flag(&ready_flags(c.base[c.rank])[p], e, c.timeout_ticks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  settle_waits(c, expired, blockIdx.x == 0, true);
  const long n4 = n >> 2;
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v[MAX_RANKS];
#pragma unroll
    for (int p = 0; p < MAX_RANKS; ++p)
      if (p < c.world) v[p] = reinterpret_cast<const f32x4*>(stage_of(c.base[p]))[i];
    f32x4 s0 = v[0];
#pragma unroll
    for (int p = 1; p < MAX_RANKS; ++p)
      if (p < c.world) s0 += v[p];
    d4[i] = s0;
  }
}

// The one-shot all-reduce fused into the optimizer launch that consumes it: after the flag exchange the Adam kernel's own
// gradient loads sum every rank's stage buffer (rank order: bit-identical everywhere), so neither the reduction kernel
// nor its result buffer exist.  Same contract as ipc_reduce_all_kernel (the publish kernel precedes it in the stream).
__global__ __launch_bounds__(256) void ipc_reduce_adam_kernel(const IpcComm c, const OptimParams p) {
  const unsigned e = *c.epoch;
  bool expired = false;
  if (threadIdx.x < c.world) {
    const int q = threadIdx.x;
    if (blockIdx.x == 0)
      __hip_atomic_store(&ready_flags(c.base[q])[c.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    expired = !wait_flag(&ready_flags(c.base[c.rank])[q], e, c.timeout_ticks);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  settle_waits(c, expired, blockIdx.x == 0, true);
  // The whole launch must take ONE apply-or-skip decision: workgroup 0 (dispatched first; the grid is co-resident) takes it
  // above and tags it with the epoch, the others wait for the tag (bounded like every wait here).
  if (c.skip) {
    unsigned* tag = c.epoch + 4;
    if (blockIdx.x == 0) {
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(tag, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (threadIdx.x == 0) {
        if (!wait_flag(tag, e, c.timeout_ticks)) __hip_atomic_store(c.skip, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
  }
  const int t = adam_tf_body<true>(p, blockIdx.x, gridDim.x, 256);
  finish_step(p, t);
}
}  // namespace

// ---- allocation / handle exchange (host)
DTF_API int dtf_ipc_alloc(long n_max_floats, void** out) {
  if (n_max_floats <= 0 || n_max_floats % 4) return -22;
  const size_t bytes = HDR_BYTES + 2 * (size_t)n_max_floats * sizeof(float);
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) return -1;
  if (hipMemset(p, 0, bytes) != hipSuccess) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  *out = p;
  return 0;
}
DTF_API int dtf_ipc_free(void* p) { return hipFree(p) == hipSuccess ? 0 : -1; }
DTF_API int dtf_ipc_get_handle(void* p, void* handle64) {
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) return -1;
  memcpy(handle64, &h, sizeof(h) <= 64 ? sizeof(h) : 64);
  return (int)sizeof(h);
}
DTF_API int dtf_ipc_open_handle(const void* handle64, void** out) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h) <= 64 ? sizeof(h) : 64);
  return hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? 0 : -1;
}
DTF_API int dtf_ipc_close_handle(void* p) { return hipIpcCloseMemHandle(p) == hipSuccess ? 0 : -1; }

DTF_API void* dtf_ipc_stage_ptr(void* base) { return reinterpret_cast<unsigned char*>(base) + HDR_BYTES; }
DTF_API void* dtf_ipc_result_ptr(void* base, long n_max) {
  return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(base) + HDR_BYTES) + n_max;
}

// Forget the expired waits counted so far: this rank's count, the copy it published to the peers and the total it has acted
// on.  Collective in effect: every rank calls it (device idle) and the callers meet at a barrier before the next exchange,
// otherwise a rank would see a total that differs from the one it acted on and skip a step.
DTF_API int dtf_ipc_reset_errors(const void* comm) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world || c.error != c.epoch + 1) return -22;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemset(c.error, 0, 4) != hipSuccess || hipMemset(c.epoch + 3, 0, 4) != hipSuccess ||
      hipMemset(c.base[c.rank] + 128, 0, 4) != hipSuccess)  // (err_pub of this rank's header)
    return -1;
  return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

// Barrier over the ranks of `comm` on the stream (one wave): bumps the epoch, tells every peer, waits for every peer.
// What it orders is up to the caller: here, "my fc1 operands for this step are in my stage buffer" (their producers
// write them there directly, stream-ordered before this kernel).
DTF_API int dtf_ipc_sync(const void* comm, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world) return -22;
  hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, c);
  DTF_CHECK_LAUNCH();
  return 0;
}

// dst[p][bytes] <- rank p's stage[0, bytes) for every peer p != rank (dst row `rank` is left alone); call after dtf_ipc_sync.
DTF_API int dtf_ipc_gather_copy(const void* comm, void* dst, long bytes, int blocks_per_peer, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (!dst || bytes <= 0 || bytes % 16 || bytes / 4 > c.n_max || c.world < 1 || c.world > MAX_RANKS) return -22;
  if (c.world == 1) return 0;
  if (blocks_per_peer < 1) blocks_per_peer = 16;
  if (blocks_per_peer > 64) blocks_per_peer = 64;
  hipLaunchKernelGGL(ipc_gather_copy_kernel, dim3((c.world - 1) * blocks_per_peer), dim3(256), 0, st, c,
                     reinterpret_cast<f32x4*>(dst), bytes / 16, blocks_per_peer);
  DTF_CHECK_LAUNCH();
  return 0;
}

// dtf_ipc_sync + dtf_ipc_gather_copy in one launch (see ipc_sync_gather_kernel).  The comm's device words must be
// [epoch, error, ticket] (ticket zero-initialised), as IpcAllReduce allocates them.
DTF_API int dtf_ipc_sync_gather(const void* comm, void* dst, long bytes, int blocks_per_peer, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (!dst || bytes <= 0 || bytes % 16 || bytes / 4 > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 ||
      c.rank >= c.world || c.error != c.epoch + 1)
    return -22;  // (device words: [epoch, error, ticket, seen, tag, spare], as IpcAllReduce allocates them)
  if (blocks_per_peer < 1) blocks_per_peer = 16;
  if (blocks_per_peer > 64) blocks_per_peer = 64;
  const int grid = c.world > 1 ? (c.world - 1) * blocks_per_peer : 1;
  hipLaunchKernelGGL(ipc_sync_gather_kernel<false>, dim3(grid), dim3(256), 0, st, c, reinterpret_cast<f32x4*>(dst), bytes / 16,
                     blocks_per_peer, GatherSlice{});
  DTF_CHECK_LAUNCH();
  return 0;
}

// dtf_ipc_sync_gather for the "shard" schedule: every rank's stage buffer holds [rows][pitch4] float4 (pool2) and, at
// tail_off4, tail_n4 float4 (dh); this rank needs columns [lo4, lo4 + w4) of the rows and the whole tail.  dst[p][bytes]
// mirrors the stage layout (only the pulled parts are written).
DTF_API int dtf_ipc_sync_gather_slice(const void* comm, void* dst, long bytes, int rows, int pitch4, int lo4, int w4,
                                      long tail_off4, long tail_n4, int blocks_per_peer, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (!dst || bytes <= 0 || bytes % 16 || bytes / 4 > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 ||
      c.rank >= c.world || c.error != c.epoch + 1)
    return -22;
  if (rows <= 0 || pitch4 <= 0 || lo4 < 0 || w4 <= 0 || lo4 + w4 > pitch4 || tail_n4 < 0 || tail_off4 < (long)rows * pitch4 ||
      (tail_off4 + tail_n4) * 16 != bytes)
    return -22;
  if (blocks_per_peer < 1) blocks_per_peer = 16;
  if (blocks_per_peer > 64) blocks_per_peer = 64;
  const int grid = c.world > 1 ? (c.world - 1) * blocks_per_peer : 1;
  const GatherSlice gs{rows, pitch4, lo4, w4, tail_off4, tail_n4, 0L};
  hipLaunchKernelGGL(ipc_sync_gather_kernel<true>, dim3(grid), dim3(256), 0, st, c, reinterpret_cast<f32x4*>(dst), 0L,
                     blocks_per_peer, gs);
  DTF_CHECK_LAUNCH();
  return 0;
}

// "Early push" form of the operand exchange ("ipc-gather2"): every rank has already stored its [rows][pitch4] block (pool2)
// into block `rank` of every peer's RESULT buffer (dtf_ipc_push_desc + shard_push_body, from spare workgroups of the fc1
// forward launch, i.e. while the fc1 forward and the loss head ran).  Here: barrier, pull only the tail (dh) from the peers'
// stage buffers, and copy the pushed rows from this rank's own result buffer into dst[p] (same layout as the stage buffer).
DTF_API int dtf_ipc_sync_gather_pushed(const void* comm, void* dst, long bytes, int rows, int pitch4, long tail_off4,
                                       long tail_n4, int blocks_per_peer, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (!dst || bytes <= 0 || bytes % 16 || bytes / 4 > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 ||
      c.rank >= c.world || c.error != c.epoch + 1)
    return -22;
  const long land4 = (long)rows * pitch4;
  if (rows <= 0 || pitch4 <= 0 || tail_n4 < 0 || tail_off4 < land4 || (tail_off4 + tail_n4) * 16 != bytes ||
      (long)c.world * land4 * 4 > c.n_max)
    return -22;
  if (blocks_per_peer < 1) blocks_per_peer = 16;
  if (blocks_per_peer > 64) blocks_per_peer = 64;
  const int grid = c.world > 1 ? (c.world - 1) * blocks_per_peer : 1;
  const GatherSlice gs{rows, pitch4, 0, pitch4, tail_off4, tail_n4, land4};
  hipLaunchKernelGGL(ipc_sync_gather_kernel<true>, dim3(grid), dim3(256), 0, st, c, reinterpret_cast<f32x4*>(dst), 0L,
                     blocks_per_peer, gs);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Descriptor of that push: `src` (n_floats, this rank's pool2 in its own stage buffer) goes to block `rank` of every rank's
// result buffer.
DTF_API int dtf_ipc_push_desc(const void* comm, float* src, long n_floats, ShardPush* push) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world || !push || !src) return -22;
  if (n_floats <= 0 || n_floats % 4 || (long)c.world * n_floats > c.n_max) return -22;
  *push = ShardPush{};
  push->src = src;
  push->n4 = n_floats / 4;
  for (int p = 0; p < c.world; ++p)
    push->dst[p] = reinterpret_cast<float*>(c.base[p] + HDR_BYTES) + c.n_max + (long)c.rank * n_floats;
  push->world = c.world;
  push->rank = c.rank;
  push->zero_src = 0;
  return 0;
}

// Descriptor of the "shard" schedule's gradient push (shard_body.h) over the communicator whose RESULT buffers are the
// landing buffers: rank r's slice [lo, lo + n) (floats, multiples of 4) of the landing buffer is written by rank r on every
// rank; `src` is the slice inside this rank's gradient arena.  dtf_ipc_sync on the same communicator, after the pushing
// launch, is the arrival barrier.
DTF_API int dtf_ipc_shard_desc(const void* comm, long lo_floats, long n_floats, float* src, int zero_src, ShardPush* push) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world || !push) return -22;
  if (lo_floats < 0 || n_floats <= 0 || lo_floats % 4 || n_floats % 4 || lo_floats + n_floats > c.n_max || !src) return -22;
  *push = ShardPush{};
  push->src = src;
  push->n4 = n_floats / 4;
  for (int p = 0; p < c.world; ++p) push->dst[p] = reinterpret_cast<float*>(c.base[p] + HDR_BYTES) + c.n_max + lo_floats;
  push->world = c.world;
  push->rank = c.rank;
  push->zero_src = zero_src;
  return 0;
}

// "wshard" schedule (engine.hip: MnistEngine::wshard): a tensor of <= n_max floats lives in BOTH buffers of every rank's allocation, stage[]
// = parity 0 and result[] = parity 1.  out[parity * 8 + r] = element `lo_floats` of rank r's buffer of that parity.
DTF_API int dtf_ipc_wshard_bufs(const void* comm, long lo_floats, long n_floats, float** out) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world || !out) return -22;
  if (lo_floats < 0 || n_floats <= 0 || lo_floats % 4 || n_floats % 4 || lo_floats + n_floats > c.n_max) return -22;
  for (int par = 0; par < 2; ++par)
    for (int r = 0; r < MAX_RANKS; ++r)
      out[par * MAX_RANKS + r] = r < c.world ? reinterpret_cast<float*>(c.base[r] + HDR_BYTES) + (par ? c.n_max : 0) + lo_floats : nullptr;
  return 0;
}

DTF_API const float* dtf_ipc_landing(const void* comm) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  return reinterpret_cast<const float*>(c.base[c.rank] + HDR_BYTES) + c.n_max;
}

// ---- general-purpose collectives over the same buffers (not on the training hot path: each is bracketed by barriers,
// so calls need no external ordering).  n counts floats (any 4-byte payload), n % 4 == 0, n <= n_max.
// dst[world][n] <- every rank's src[n]
DTF_API int dtf_ipc_all_gather(const void* comm, const float* src, float* dst, long n, int blocks, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (!src || !dst || n <= 0 || n % 4 || n > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world)
    return -22;
  if (blocks < 1) blocks = 32;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(ipc_publish_kernel, dim3(blocks), dim3(256), 0, st, c, src, n);
  hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, c);
  for (int p = 0; p < c.world; ++p)
    hipLaunchKernelGGL(ipc_copy_from_kernel, dim3(blocks), dim3(256), 0, st, c, p, reinterpret_cast<f32x4*>(dst + (long)p * n),
                       n / 4);
  hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, c);  // everyone has read: stage buffers may be reused
  DTF_CHECK_LAUNCH();
  return 0;
}

// buf[n] on every rank <- rank root's buf[n]
DTF_API int dtf_ipc_broadcast(const void* comm, float* buf, long n, int root, int blocks, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (!buf || n <= 0 || n % 4 || n > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world ||
      root < 0 || root >= c.world)
    return -22;
  if (blocks < 1) blocks = 32;
  if (blocks > 256) blocks = 256;
  // (every rank runs the publish kernel so that the epochs stay in step; only the root has something to copy)
  hipLaunchKernelGGL(ipc_publish_kernel, dim3(c.rank == root ? blocks : 1), dim3(256), 0, st, c, buf, c.rank == root ? n : 0L);
  hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, c);
  if (c.rank != root)
    hipLaunchKernelGGL(ipc_copy_from_kernel, dim3(blocks), dim3(256), 0, st, c, root, reinterpret_cast<f32x4*>(buf), n / 4);
  hipLaunchKernelGGL(ipc_sync_kernel, dim3(1), dim3(64), 0, st, c);
  DTF_CHECK_LAUNCH();
  return 0;
}

// dst <- sum over ranks of src (n floats, n % 4 == 0), two kernels.  CONTRACT: a rank overwrites its stage buffer at its
// next call, so between two calls every rank must pass some other barrier over the same ranks (in the gather schedule:
// the operand exchange's dtf_ipc_sync at the start of the next step, which no rank reaches before this reduction of
// the previous step has finished on it).
DTF_API int dtf_ipc_allreduce_oneshot(const void* comm, const float* src, float* dst, long n, int blocks, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (n <= 0 || n % 4 || n > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world) return -22;
  if (blocks < 1) blocks = 32;
  const long want = (n / 4 + 255) / 256;
  if (blocks > want) blocks = (int)want;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(ipc_publish_kernel, dim3(blocks), dim3(256), 0, st, c, src, n);
  hipLaunchKernelGGL(ipc_reduce_all_kernel, dim3(blocks), dim3(256), 0, st, c, dst, n);
  DTF_CHECK_LAUNCH();
  return 0;
}

// First half of dtf_ipc_allreduce_oneshot on its own: copy src into this rank's stage buffer (the flag exchange and
// the reduction happen in the consumer, dtf_ipc_adam_oneshot).
DTF_API int dtf_ipc_publish(const void* comm, const float* src, long n, int blocks, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (!src || n <= 0 || n % 4 || n > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world) return -22;
  if (blocks < 1) blocks = 32;
  const long want = (n / 4 + 256 * COPY_U - 1) / (256 * COPY_U);
  if (blocks > want) blocks = (int)want;
  hipLaunchKernelGGL(ipc_publish_kernel, dim3(blocks), dim3(256), 0, st, c, src, n);
  DTF_CHECK_LAUNCH();
  return 0;
}

// TF-Adam over op's range with the gradient taken as the sum of every rank's published buffer (op->n floats each, see
// dtf_ipc_publish): op->G is only re-zeroed, op->Gsrc / Gsrcs are set here.  Same "another barrier before the next
// publish" contract as dtf_ipc_allreduce_oneshot.
DTF_API int dtf_ipc_adam_oneshot(const void* comm, const OptimParams* op, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world) return -22;
  if (!op || op->n <= 0 || op->n % 4 || op->n > c.n_max || op->zero_from % 4 || !op->step) return -22;
  OptimParams p = *op;
  p.Gsrc = nullptr;
  p.nsrc = c.world;
  for (int r = 0; r < MAX_RANKS; ++r)
    p.Gsrcs[r] = r < c.world ? reinterpret_cast<const float*>(c.base[r] + HDR_BYTES) : nullptr;
  long blocks = (p.n / 4 + 1023) / 1024;  // 4 float4 per lane per pass, as the plain optimizer launch
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ipc_reduce_adam_kernel, dim3((int)blocks), dim3(256), 0, st, c, p);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Zero-copy all-reduce of n floats: stage[0, rest_off) of this rank already holds its data, src_rest[0, rest_n) is
// copied to stage[rest_off, ...) (rest_off + rest_n == n); afterwards result[0, n) of this rank holds the sum.
DTF_API int dtf_ipc_allreduce_zc(const void* comm, const float* src_rest, long rest_off, long rest_n, long n, int blocks,
                                 hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (n <= 0 || n % 4 || n > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world) return -22;
  if (rest_off < 0 || rest_n < 0 || rest_off % 4 || rest_n % 4 || rest_off + rest_n != n) return -22;
  if (blocks < 1) blocks = 64;
  if (blocks > 1024) blocks = 1024;
  const long rest_blocks = (rest_n / 4 + 256 * COPY_U - 1) / (256 * COPY_U);
  hipLaunchKernelGGL(ipc_publish_rest_kernel, dim3(rest_blocks < 1 ? 1 : (rest_blocks > blocks ? blocks : (int)rest_blocks)),
                     dim3(256), 0, st, c, src_rest, rest_off, rest_n);
  hipLaunchKernelGGL(ipc_reduce_kernel, dim3(blocks), dim3(256), 0, st, c, n);
  hipLaunchKernelGGL(ipc_done_kernel, dim3(1), dim3(64), 0, st, c);
  DTF_CHECK_LAUNCH();
  return 0;
}

// comm: struct IpcComm by pointer (host memory).  src/dst may alias (in-place all-reduce of a gradient arena).
DTF_API int dtf_ipc_allreduce(const void* comm, const float* src, float* dst, long n, int blocks, hipStream_t st) {
  const IpcComm& c = *reinterpret_cast<const IpcComm*>(comm);
  if (n <= 0 || n % 4 || n > c.n_max || c.world < 1 || c.world > MAX_RANKS || c.rank < 0 || c.rank >= c.world) return -22;
  if (blocks < 1) blocks = 64;
  // (ranks that share one GPU must keep blocks <= 128, half the CUs, so that both ranks' spinning grids stay co-resident;
  //  the caller knows the placement, see DataParallelTrainer._try_ipc)
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(ipc_publish_kernel, dim3(blocks), dim3(256), 0, st, c, src, n);
  hipLaunchKernelGGL(ipc_reduce_kernel, dim3(blocks), dim3(256), 0, st, c, n);
  hipLaunchKernelGGL(ipc_collect_kernel, dim3(blocks), dim3(256), 0, st, c, dst, n);
  DTF_CHECK_LAUNCH();
  return 0;
}
