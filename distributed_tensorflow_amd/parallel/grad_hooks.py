"""Gradient hooks of the native MNIST step: what the engine calls after the fc gradients (part 1) and after all gradients
(part 2) of a step have been enqueued, so that the exchange becomes part of the step's hipGraph.  One class per transport:
the IPC exchange buffers (``ipc_comm.py``: staged, overlapped, zero-copy and operand-gather forms) and the native RCCL
communicator (``rccl_comm.py``).  Reference behaviour being replaced: every worker pushes its gradients to the parameter
server each step (``/root/reference/demo2/train.py:27-29,146-149``)."""
from __future__ import annotations

import os

import torch


class IpcGradHook:
    """Gradient all-reduce of the native MNIST step over the IPC transport.  With ``overlap`` the fc bucket (12.9 MB,
    final after step part 1) is reduced on a forked stream while the conv backward (part 2) runs on the main one —
    event fork/join, so it also becomes a parallel branch of the captured hipGraph — and only the 0.2 MB conv bucket
    is reduced after it."""

    def __init__(self, ipc, bucket_a: torch.Tensor, bucket_b: torch.Tensor, whole: torch.Tensor, overlap: bool):
        self.ipc, self.a, self.b, self.whole, self.overlap = ipc, bucket_a, bucket_b, whole, overlap
        # zero-copy mode (set_zero_copy): the fc1 filter gradient, arena [0, zc_off), is produced straight into the
        # exchange buffer and the optimizer reads the reduced arena from the result buffer; see ipc_allreduce.hip
        self.zc_off = 0
        # gather mode (set_gather): the fc1 filter gradient, arena [0, rest_off), is computed globally from gathered
        # operands inside the step; only the rest of the arena is all-reduced
        self.rest_off = 0
        # (the engine's DTF_EARLY_FC1_ADAM=0 debugging switch applies the whole arena in one optimizer launch, which then
        # needs the rest of the arena reduced in place as before)
        self.fused_apply = os.environ.get("DTF_EARLY_FC1_ADAM", "1") != "0"
        if overlap:
            self.side = torch.cuda.Stream(device=whole.device)
            self.ev_fork = torch.cuda.Event()
            self.ev_join = torch.cuda.Event()

    def set_overlap(self, overlap: bool) -> None:
        self.overlap = overlap
        if overlap and not hasattr(self, "side"):
            self.side = torch.cuda.Stream(device=self.whole.device)
            self.ev_fork = torch.cuda.Event()
            self.ev_join = torch.cuda.Event()

    def set_zero_copy(self, rest_off: int) -> None:
        self.zc_off = rest_off
        if rest_off:
            self.overlap = False

    def set_gather(self, rest_off: int) -> None:
        self.rest_off = rest_off
        if rest_off:
            self.overlap = False

    @property
    def apply_comm(self):
        """Non-None in gather mode: the engine's optimizer launch for the arena after W_fc1 must sum the ranks' published
        gradients itself (``dtf_mnist_apply_part_ipc`` on this comm struct)."""
        return self.ipc.comm if self.rest_off and self.fused_apply else None

    def after_part1(self, stream) -> None:
        if not self.overlap:
            return
        self.ev_fork.record(stream)
        self.side.wait_event(self.ev_fork)
        self.ipc.all_reduce(self.a, stream=self.side.cuda_stream)
        self.ev_join.record(self.side)

    def after_part2(self, stream) -> None:
        if self.rest_off:
            # one-shot form (its "next call only after another barrier" contract is met by the operand exchange's
            # barrier after the next step's loss head), with the reduction fused into the optimizer launch: only the
            # copy into the exchange buffer happens here, see apply_comm
            if self.fused_apply:
                self.ipc.publish(self.whole[self.rest_off:], stream=stream.cuda_stream)
            else:
                self.ipc.all_reduce_oneshot(self.whole[self.rest_off:], stream=stream.cuda_stream)
            return
        if self.zc_off:
            self.ipc.all_reduce_zero_copy(self.whole[self.zc_off:], self.zc_off, self.whole.numel(),
                                          stream=stream.cuda_stream)
            return
        if not self.overlap:
            self.ipc.all_reduce(self.whole, stream=stream.cuda_stream)
            return
        stream.wait_event(self.ev_join)
        self.ipc.all_reduce(self.b, stream=stream.cuda_stream)


class RcclGradHook:
    """Gradient all-reduce of the native MNIST step over the native RCCL communicator (``parallel/rccl_comm.py``): the
    ncclAllReduce calls are enqueued on the engine's stream(s) inside the step, so they are captured into the step's
    hipGraph and the whole data-parallel step is one graph launch.  With ``overlap`` the fc bucket (12.9 MB, final after
    step part 1) is reduced on a forked stream while the conv backward runs on the main one, as in ``IpcGradHook``."""

    apply_comm = None

    def __init__(self, rccl, bucket_a: torch.Tensor, bucket_b: torch.Tensor, whole: torch.Tensor, overlap: bool):
        self.rccl, self.a, self.b, self.whole, self.overlap = rccl, bucket_a, bucket_b, whole, overlap
        if overlap:
            self.side = torch.cuda.Stream(device=whole.device)
            self.ev_fork = torch.cuda.Event()
            self.ev_join = torch.cuda.Event()

    def after_part1(self, stream) -> None:
        if not self.overlap:
            return
        self.ev_fork.record(stream)
        self.side.wait_event(self.ev_fork)
        self.rccl.all_reduce(self.a, stream=self.side)
        self.ev_join.record(self.side)

    def after_part2(self, stream) -> None:
        if not self.overlap:
            self.rccl.all_reduce(self.whole, stream=stream)
            return
        stream.wait_event(self.ev_join)
        self.rccl.all_reduce(self.b, stream=stream)
