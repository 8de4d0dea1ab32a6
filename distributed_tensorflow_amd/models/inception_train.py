"""Inception-v3 full fine-tune (BASELINE config 5: "Inception-v3 full fine-tune bf16, DP") — training executor.

The reference only ever trains the 2048->C head on frozen bottlenecks (``retrain1/retrain.py:262-297``); SURVEY §2.6
lists the full fine-tune as the scale-up configuration: every conv gets training-mode BatchNorm forward/backward,
backward-data and filter gradients, and the ≈95 MB of fp32 gradients are all-reduced across GPUs.

Design (MI355X-first):

* one flat fp32 arena ``P`` (and ``G``; ``M``/``V`` for Adam) holds the head and, per conv, ``W`` (HWIO, TensorFlow's
  layout), ``gamma``, ``beta`` — ordered by *gradient readiness* (head first, then convs in reverse execution order)
  so contiguous arena ranges become final while the backward is still running and can be all-reduced bucket by
  bucket on RCCL's stream (xGMI is per-link bound: few large buckets);
* activations ``y`` (bf16 NHWC, branches written straight into their concat slice), raw conv outputs ``z`` (bf16)
  and activation gradients (bf16) are preallocated per batch size — ≈57 MB/image, i.e. B=256 fits many times over in
  288 GB of HBM — and ``z`` is overwritten in place by ``dz`` in the backward;
* per conv: forward = MFMA implicit GEMM whose epilogue also accumulates the BN statistics, a 1-block finalize, one
  fused normalise+ReLU pass; backward = BN reduce, BN apply (in place), filter-gradient GEMM, data-gradient GEMM
  (``csrc/hip/conv_train.hip``);
* the fp32 master weights are read directly by the GEMM loaders (converted to bf16 while staged in LDS): the
  optimizer touches one arena, no bf16 shadow copy exists;
* the tiny head (``[B,2048]x[2048,C]``) uses torch matmuls on arena views with the closed-form (double-)softmax
  gradient.

``torch_train_step`` is the PyTorch autograd oracle of exactly the same computation (used by the tests and as the
eager baseline in ``scripts/bench_inception_train.py``).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .inception_v3 import BN_EPS, BOTTLENECK_TENSOR_SIZE, INPUT_PAD_CHANNELS, InceptionV3, build_graph

ALIGN = 64  # arena segment alignment in elements (256 B)
STAT_SLOTS = 16  # DTF_STAT_SLOTS in csrc/hip/gemm_core.h


def _round_up(n: int, a: int = ALIGN) -> int:
    return (n + a - 1) // a * a


class ArenaLayout:
    """Offsets of every trainable tensor in the flat arena, in gradient-readiness order."""

    def __init__(self, graph, num_classes: int):
        self.entries: "Dict[str, Tuple[int, Tuple[int, ...]]]" = {}
        off = 0

        def add(name, shape):
            nonlocal off
            n = 1
            for d in shape:
                n *= d
            self.entries[name] = (off, tuple(shape))
            off += _round_up(n)

        add("final_training_ops/weights/final_weights", (BOTTLENECK_TENSOR_SIZE, num_classes))
        add("final_training_ops/biases/final_biases", (num_classes,))
        self.conv_order: List[str] = [op.conv.name for op in reversed(graph.ops) if op.kind == "conv"]
        self.conv_end: Dict[str, int] = {}  # arena offset at which conv `name`'s segment ends
        for name in self.conv_order:
            s = graph.convs[name]
            add(name + "/conv2d_params", (s.kh, s.kw, s.cin, s.cout))
            add(name + "/batchnorm/gamma", (s.cout,))
            add(name + "/batchnorm/beta", (s.cout,))
            self.conv_end[name] = off
        self.size = off

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.entries[name]
        n = 1
        for d in shape:
            n *= d
        return flat[off:off + n].view(shape)


def softmax_xent_grad(logits: torch.Tensor, labels: torch.Tensor, double_softmax: bool):
    """(mean loss, dloss/dlogits, probabilities) for CE(softmax(z)) or the reference's CE(softmax(softmax(z)))
    (``retrain1/retrain.py:278,282``)."""
    B, C = logits.shape
    p = torch.softmax(logits, 1)
    onehot = F.one_hot(labels.long(), C).to(logits.dtype)
    if not double_softmax:
        loss = -(torch.log_softmax(logits, 1) * onehot).sum(1).mean()
        return loss, (p - onehot) / B, p
    loss = -(torch.log_softmax(p, 1) * onehot).sum(1).mean()
    dp = (torch.softmax(p, 1) - onehot) / B
    return loss, p * (dp - (dp * p).sum(1, keepdim=True)), p


class InceptionTrainer:
    """Native full-network training step for a fixed batch size."""

    def __init__(self, num_classes: int, batch_size: int, device="cuda", *, lr: float = 0.01, optimizer: str = "sgd",
                 seed: int = 0, init: Optional[InceptionV3] = None, bn_momentum: float = 0.9997,
                 double_softmax: bool = False, head_std: float = 0.001, conv_core: str = "auto",
                 async_lds: Optional[bool] = None):
        from .. import native
        self.lib = native.hip()
        # conv GEMM core per layer and pass: "v1" = register-staged 128x64 tiles (gemm_core.h: small LDS footprint, 4-5
        # workgroups per CU hide the short K loops of this network), "v2" = async global->LDS ring with 256-row tiles
        # (igemm2.hip: wins where one M tile has many K-steps and the grid is large), "auto" = time both once per layer
        # at start-up and keep the faster (measure, don't guess: at B=64 v2 wins 20-30 of the 281 kernels).
        if async_lds is not None:
            conv_core = "v2" if async_lds else "v1"
        if conv_core not in ("v1", "v2", "auto"):
            raise ValueError("conv_core must be v1, v2 or auto")
        self.conv_core = conv_core
        self.async_lds = conv_core != "v1"
        self.native = native
        self.device = torch.device(device)
        self.B = batch_size
        self.C = num_classes
        self.lr = lr
        self.optimizer = optimizer
        self.bn_momentum = bn_momentum
        self.double_softmax = double_softmax
        self.world = 1
        self.graph = build_graph()
        g = self.graph
        self.layout = ArenaLayout(g, num_classes)
        dev = self.device
        self.P = torch.zeros(self.layout.size, dtype=torch.float32, device=dev)
        self.G = torch.zeros_like(self.P)
        if optimizer == "adam":
            self.M = torch.zeros_like(self.P)
            self.V = torch.zeros_like(self.P)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        src = init if init is not None else InceptionV3(device="cpu", seed=seed)
        gen = torch.Generator(device="cpu").manual_seed(seed + 17)
        hw = torch.empty(BOTTLENECK_TENSOR_SIZE, num_classes)
        torch.nn.init.trunc_normal_(hw, std=head_std, a=-2 * head_std, b=2 * head_std, generator=gen)
        self.param("final_training_ops/weights/final_weights").copy_(hw)
        # moving statistics (not trained): [mean | var] per conv in one buffer
        self.stat_off: Dict[str, int] = {}
        off = 0
        for name, s in g.convs.items():
            self.stat_off[name] = off
            off += _round_up(s.cout, 8)
        self.stat_total = off
        self.moving_mean = torch.zeros(off, dtype=torch.float32, device=dev)
        self.moving_var = torch.ones(off, dtype=torch.float32, device=dev)
        for name, s in g.convs.items():
            p = src.params[name]
            w = p["weights"]
            if w.shape[2] != s.cin:  # first conv: 3 -> 8 zero-padded input channels
                w = torch.cat([w, torch.zeros(s.kh, s.kw, s.cin - w.shape[2], s.cout)], dim=2)
            self.param(name + "/conv2d_params").copy_(w)
            self.param(name + "/batchnorm/gamma").copy_(p["gamma"])
            self.param(name + "/batchnorm/beta").copy_(p["beta"])
            o = self.stat_off[name]
            self.moving_mean[o:o + s.cout].copy_(p["moving_mean"])
            self.moving_var[o:o + s.cout].copy_(p["moving_variance"])
        self._fwd_set = 0
        # filter gradients on a second stream (``backward``): 0 switches it off
        self.dw_side = os.environ.get("DTF_FT_DW_STREAM", "1") != "0"
        self._dw_stream = None
        # forward: independent branches of a mixed block on up to this many streams (``_plan_streams``); 1 = one stream
        # (measured at batch 128, filter gradients already on their stream: 19.95 / 19.2 / 19.6 ms per step with 1 / 2 / 3; from
        # three streams on, edges between side streams have to be relayed through the main one, see StreamPlan)
        self.fwd_streams = max(1, int(os.environ.get("DTF_FT_FWD_STREAMS", "2")))
        self._fwd_plan = None
        self._alloc()
        self._plan_backward()
        pick = 2 if conv_core == "v2" else 1  # (the autotuner also considers 3 = the 128-row-tile variant of v2)
        self.core: Dict[str, Dict[str, int]] = {name: {"fwd": pick, "bwd": pick, "dw": pick} for name in g.convs}
        if self.async_lds:
            self._alloc_packs()
        if conv_core == "auto":
            self._autotune()

    def _alloc_packs(self) -> None:
        """bf16 K-contiguous packings of every conv weight for the async-LDS GEMM core: Wp [Cout][tap*Cin] (forward) and
        Wq [Cin][tap*Cout] (backward-data), refreshed from the fp32 arena by one kernel launch per step."""
        import ctypes as C

        class PackDesc(C.Structure):  # csrc/hip/igemm2.hip: struct PackDesc
            _fields_ = [("w_off", C.c_long), ("p_off", C.c_long), ("q_off", C.c_long), ("taps", C.c_int), ("cin", C.c_int),
                        ("cout", C.c_int), ("pad", C.c_int)]
        self.native.check(self.lib.dtf_igemm2_init() or self.lib.dtf_igemm2_dw_init(), "igemm2_init")
        descs = (PackDesc * len(self.graph.convs))()
        self.pack_off: Dict[str, int] = {}
        off = 0
        for k, (name, s) in enumerate(self.graph.convs.items()):
            n = s.kh * s.kw * s.cin * s.cout
            descs[k] = PackDesc(self.layout.entries[name + "/conv2d_params"][0], off, off, s.kh * s.kw, s.cin, s.cout, 0)
            self.pack_off[name] = off
            off += _round_up(n)
        self.Wp = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.Wq = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self._pack_descs = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)

    def pack_weights(self) -> None:
        st = torch.cuda.current_stream().cuda_stream
        self.native.check(self.lib.dtf_pack_conv_weights(self.P.data_ptr(), self._pack_descs.data_ptr(), len(self.graph.convs),
                                                         self.Wp.data_ptr(), self.Wq.data_ptr(), st), "pack_conv_weights")

    # ------------------------------------------------------------------ arena
    def param(self, name: str) -> torch.Tensor:
        return self.layout.view(self.P, name)

    def grad(self, name: str) -> torch.Tensor:
        return self.layout.view(self.G, name)

    @property
    def num_params(self) -> int:
        n = 0
        for _name, (_off, shape) in self.layout.entries.items():
            k = 1
            for d in shape:
                k *= d
            n += k
        return n

    def export_to(self, model: InceptionV3) -> None:
        """Copy trained weights + moving statistics into an inference ``InceptionV3`` (folded-BN path)."""
        layers = {}
        for name, s in self.graph.convs.items():
            w = self.param(name + "/conv2d_params").detach().cpu()
            o = self.stat_off[name]
            layers[name] = {"weights": w[:, :, :model.params[name]["weights"].shape[2], :].contiguous().numpy(),
                            "gamma": self.param(name + "/batchnorm/gamma").cpu().numpy(),
                            "beta": self.param(name + "/batchnorm/beta").cpu().numpy(),
                            "moving_mean": self.moving_mean[o:o + s.cout].cpu().numpy(),
                            "moving_variance": self.moving_var[o:o + s.cout].cpu().numpy()}
        model.load_tf_weights(layers)

    # ------------------------------------------------------------------ buffers
    def _alloc(self) -> None:
        g, B, dev = self.graph, self.B, self.device
        bf = torch.bfloat16
        self.act: Dict[str, torch.Tensor] = {}
        self.dact: Dict[str, torch.Tensor] = {}
        for name, (h, w, c) in g.shapes.items():
            if name == "input":
                continue
            self.act[name] = torch.zeros(B, h, w, c, dtype=bf, device=dev)
            self.dact[name] = torch.zeros(B, h, w, c, dtype=bf, device=dev)
        self.z: Dict[str, torch.Tensor] = {}
        self.argmax: Dict[int, torch.Tensor] = {}
        for i, op in enumerate(g.ops):
            if op.kind == "conv":
                ho, wo, _ = g.shapes[op.dst]
                self.z[op.conv.name] = torch.empty(B, ho, wo, op.conv.cout, dtype=bf, device=dev)
            elif op.kind == "maxpool":
                ho, wo, _ = g.shapes[op.dst]
                self.argmax[i] = torch.empty(B, ho, wo, op.channels, dtype=torch.uint8, device=dev)
        # statistic accumulators: STAT_SLOTS replicas of [fwd sum|sumsq] and of [bwd s1|s2] per conv (their consumer
        # kernels re-zero them, so no per-step memset), the folded backward sums, and one ticket per conv
        # (blocks 0 and 3 = the two alternating forward sets; blocks 1 and 2 = the backward ones)
        self.scratch = torch.zeros(4 * STAT_SLOTS * 2 * self.stat_total, dtype=torch.float32, device=dev)
        self._bwd_parity = 0
        self._fwd_parity = 0
        self.bn = torch.zeros(4 * self.stat_total, dtype=torch.float32, device=dev)  # mean, rstd, scale, shift per conv
        self.feat = torch.zeros(B, BOTTLENECK_TENSOR_SIZE, dtype=torch.float32, device=dev)
        self.dfeat = torch.zeros_like(self.feat)
        self._last_metrics = torch.zeros(2, dtype=torch.float32, device=dev)  # (loss sum, correct count) of the last step

    def activation_bytes(self) -> int:
        n = sum(t.numel() * t.element_size() for d in (self.act, self.dact, self.z) for t in d.values())
        return n + sum(t.numel() for t in self.argmax.values())

    def _plan_backward(self) -> None:
        """Static first-writer analysis: the first backward op that produces a gradient for a buffer stores, later
        ones (other branches reading the same tensor) accumulate."""
        written = set()
        self.accumulate: Dict[int, bool] = {}
        ops = self.graph.ops
        for i in range(len(ops) - 1, -1, -1):
            op = ops[i]
            if op.kind == "gap":
                written.add(op.src)
                continue
            if op.src == "input":
                continue
            self.accumulate[i] = op.src in written
            written.add(op.src)

    # ------------------------------------------------------------------ forward / backward
    def _stats(self, name: str, which: int) -> int:
        """Device pointer of the [STAT_SLOTS][2][Cout] accumulator block of conv ``name`` (0 / 3 = forward statistics of
        even / odd steps, 1 / 2 = the backward sums of even / odd steps)."""
        return self.scratch.data_ptr() + 4 * STAT_SLOTS * 2 * (self.stat_total * which + self.stat_off[name])

    def _bn_ptr(self, name: str) -> int:
        return self.bn.data_ptr() + 4 * 4 * self.stat_off[name]

    # ------------------------------------------------------------------ conv kernels (core 1 = gemm_core.h, 2 = igemm2.hip)
    def _conv_fwd(self, op, src, core: int, st: int) -> None:
        s, lib = op.conv, self.lib
        B, H, W, Cs = src.shape
        _, Ho, Wo, _ = self.act[op.dst].shape
        z = self.z[s.name]
        if core >= 2:  # 2..5 = async-LDS core, variant = core - 2 (bit 0: 128-row tiles, bit 1: 64-column tiles)
            rc = lib.dtf_conv_fwd_stats_v2(src.data_ptr(), self.Wp.data_ptr() + 2 * self.pack_off[s.name], z.data_ptr(),
                                           self._stats(s.name, self._fwd_set), B, H, W, s.cin, Cs, s.cout, s.kh, s.kw, s.stride,
                                           s.pad[0], s.pad[1], Ho, Wo, core - 2, st)
        else:
            rc = lib.dtf_conv_fwd_stats(src.data_ptr(), self.param(s.name + "/conv2d_params").data_ptr(), z.data_ptr(),
                                        self._stats(s.name, self._fwd_set), B, H, W, s.cin, Cs, s.cout, s.kh, s.kw, s.stride,
                                        s.pad[0], s.pad[1], Ho, Wo, st)
        self.native.check(rc, "conv_fwd_stats(core %d) %s" % (core, s.name))

    def _conv_dw(self, op, src, core: int, st: int) -> None:
        s, lib = op.conv, self.lib
        B, H, W, Cs = src.shape
        _, Ho, Wo, _ = self.act[op.dst].shape
        z, dW = self.z[s.name], self.grad(s.name + "/conv2d_params")
        if core >= 2:  # 2 = 128-wide tiles when both channel counts exceed 64, 3 = 64-wide tiles
            rc = lib.dtf_conv_dw_v2(src.data_ptr(), Cs, z.data_ptr(), dW.data_ptr(), B, H, W, s.cin, s.cout, s.kh, s.kw,
                                    s.stride, s.pad[0], s.pad[1], Ho, Wo, self._dw2_splits(s, B * Ho * Wo, core - 2), core - 2, st)
        else:
            rc = lib.dtf_conv_dw_general(src.data_ptr(), Cs, z.data_ptr(), dW.data_ptr(), B, H, W, s.cin, s.cout, s.kh, s.kw,
                                         s.stride, s.pad[0], s.pad[1], Ho, Wo,
                                         self._dw_splits(s.kh * s.kw * s.cin, s.cout, B * Ho * Wo), st)
        self.native.check(rc, "conv_dw(core %d) %s" % (core, s.name))

    def _conv_bwd_data(self, op, src, core: int, accumulate: int, st: int) -> None:
        s, lib = op.conv, self.lib
        B, H, W, Cs = src.shape
        _, Ho, Wo, _ = self.act[op.dst].shape
        z, dx = self.z[s.name], self.dact[op.src]
        if core >= 2:
            rc = lib.dtf_conv_bwd_data_v2(z.data_ptr(), self.Wq.data_ptr() + 2 * self.pack_off[s.name], dx.data_ptr(), B, H, W,
                                          s.cin, Cs, s.cout, s.kh, s.kw, s.stride, s.pad[0], s.pad[1], Ho, Wo, accumulate,
                                          core - 2, st)
        else:
            rc = lib.dtf_conv_bwd_data(z.data_ptr(), self.param(s.name + "/conv2d_params").data_ptr(), dx.data_ptr(), B, H, W,
                                       s.cin, Cs, s.cout, s.kh, s.kw, s.stride, s.pad[0], s.pad[1], Ho, Wo, accumulate, st)
        self.native.check(rc, "conv_bwd_data(core %d) %s" % (core, s.name))

    def _autotune(self, reps: int = 3) -> None:
        """Time both conv cores on every layer's real buffers (contents irrelevant) and keep the faster per pass.
        Leaves the accumulators it touched (BN statistic slots, gradient arena) zeroed again."""
        st = torch.cuda.current_stream().cuda_stream
        x = torch.zeros(self.B, 299, 299, INPUT_PAD_CHANNELS, dtype=torch.bfloat16, device=self.device)
        self.pack_weights()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

        def timed(fn) -> float:
            fn()  # warm-up (code object load)
            best = float("inf")
            for _ in range(reps):
                ev[0].record()
                fn()
                ev[1].record()
                ev[1].synchronize()
                best = min(best, ev[0].elapsed_time(ev[1]))
            return best

        self.tuning: Dict[str, Dict[str, Dict[int, float]]] = {}
        for op in self.graph.ops:
            if op.kind != "conv":
                continue
            src = x if op.src == "input" else self.act[op.src]
            name = op.conv.name
            res = {}
            s_ = op.conv
            wide_n = (2, 3, 4, 5) if s_.cout > 64 else (2, 3)  # the 64-column variants only differ when N > 64
            wide_k = (2, 3, 4, 5) if s_.cin > 64 else (2, 3)
            res["fwd"] = {c: timed(lambda c=c: self._conv_fwd(op, src, c, st)) for c in (1,) + wide_n}
            res["dw"] = {c: timed(lambda c=c: self._conv_dw(op, src, c, st))
                         for c in ((1, 2, 3) if (s_.cin > 64 and s_.cout > 64) else (1, 2))}
            if op.src != "input":
                # (6..9 = the stride-2 layers by parity class: igemm2.hip variant bit 2 on top of the tile bits)
                by_class = ((6, 7, 8, 9) if s_.cin > 64 else (6, 7)) if s_.stride == 2 else ()
                res["bwd"] = {c: timed(lambda c=c: self._conv_bwd_data(op, src, c, 0, st)) for c in (1,) + wide_k + by_class}
            self.tuning[name] = res
            for k, ts in res.items():
                best = min(ts, key=ts.get)
                self.core[name][k] = best if ts[best] < 0.97 * ts[1] else 1
        torch.cuda.synchronize()
        self.scratch.zero_()
        self.G.zero_()

    def tuning_summary(self) -> str:
        if not getattr(self, "tuning", None):
            return "conv core: %s everywhere" % self.conv_core
        out = []
        for k in ("fwd", "bwd", "dw"):
            rows = {name: r[k] for name, r in self.tuning.items() if k in r}
            counts: Dict[int, int] = {}
            for name in rows:
                counts[self.core[name][k]] = counts.get(self.core[name][k], 0) + 1
            out.append("%s: layers per core %s, core 1 everywhere %.2f ms, chosen %.2f ms" % (
                k, dict(sorted(counts.items())), sum(r[1] for r in rows.values()),
                sum(r[self.core[name][k]] for name, r in rows.items())))
        return "; ".join(out)

    def _op_cost_us(self, op, which: str = "fwd") -> float:
        """Estimated duration of one op of the program, for the stream planner: the autotuner's measurement where there is
        one, else FLOPs / bytes at typical rates of this network."""
        B = self.B
        if op.kind == "conv":
            s = op.conv
            _, Ho, Wo, _ = self.act[op.dst].shape
            t = getattr(self, "tuning", {}).get(s.name, {}).get(which)
            us = 1e3 * t[self.core[s.name][which]] if t else 2.0 * B * Ho * Wo * s.cout * s.kh * s.kw * s.cin / 300e12 * 1e6
            return us + 4.0 + 6.0 * B * Ho * Wo * s.cout / 3e12 * 1e6    # + the batch-norm pass over the output
        src = self.act[op.src] if op.src != "input" else None
        return 4.0 + (4.0 * src.numel() / 3e12 * 1e6 if src is not None else 0.0)

    def forward(self, x: torch.Tensor, training: bool = True) -> torch.Tensor:
        g, B, lib, chk = self.graph, self.B, self.lib, self.native.check
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and tuple(x.shape) == (B, 299, 299, INPUT_PAD_CHANNELS)
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        self._x = x
        if self.async_lds:
            self.pack_weights()
        fcur, fprev = (3, 0) if self._fwd_parity else (0, 3)
        self._fwd_parity ^= 1
        self._fwd_set = fcur
        plan = None
        if self.fwd_streams > 1:
            if self._fwd_plan is None:
                from .inception_v3 import StreamPlan
                self._fwd_plan = StreamPlan(g.ops, [self._op_cost_us(op) for op in g.ops], self.fwd_streams, self.device)
            plan = self._fwd_plan
            plan.begin(main)
        for i, op in enumerate(g.ops):
            src = x if op.src == "input" else self.act[op.src]
            _, H, W, Cs = src.shape
            if plan is not None:
                st = plan.stream_for(i).cuda_stream
            if op.kind == "conv":
                s = op.conv
                dst = self.act[op.dst]
                _, Ho, Wo, Cd = dst.shape
                z = self.z[s.name]
                M = B * Ho * Wo
                self._conv_fwd(op, src, self.core[s.name]["fwd"], st)
                o = self.stat_off[s.name]
                chk(lib.dtf_bn_finalize_relu_fwd(self._stats(s.name, fcur), self._stats(s.name, fprev),
                                                 self.param(s.name + "/batchnorm/gamma").data_ptr(),
                                                 self.param(s.name + "/batchnorm/beta").data_ptr(),
                                                 self.moving_mean.data_ptr() + 4 * o if training else 0,
                                                 self.moving_var.data_ptr() + 4 * o if training else 0,
                                                 self._bn_ptr(s.name), z.data_ptr(), dst.data_ptr() + 2 * op.dst_off, M, s.cout, Cd,
                                                 BN_EPS, self.bn_momentum, st), "bn_finalize_relu_fwd")
            elif op.kind == "maxpool":
                dst = self.act[op.dst]
                chk(lib.dtf_maxpool3x3_arg(src.data_ptr(), dst.data_ptr() + 2 * op.dst_off, self.argmax[i].data_ptr(), B, H, W,
                                           op.channels, Cs, dst.shape[3], op.stride, st), "maxpool_arg")
            elif op.kind == "avgpool":
                dst = self.act[op.dst]
                chk(lib.dtf_avgpool3x3_same(src.data_ptr(), dst.data_ptr(), B, H, W, op.channels, Cs, dst.shape[3], st), "avgpool")
            elif op.kind == "gap":
                chk(lib.dtf_global_avgpool(src.data_ptr(), self.feat.data_ptr(), B, H * W, Cs, st), "gap")
            if plan is not None:
                plan.done(i)
        if plan is not None:
            plan.end()
        return self.feat

    def head_forward_backward(self, labels: torch.Tensor) -> None:
        """The new softmax layer on the pooled features, through the retrain-head kernels (``csrc/hip/retrain_head.hip``):
        logits + (double-)softmax cross-entropy + accuracy + dlogits in one launch, the filter / bias gradient in a
        second, the gradient w.r.t. the features in a third — no library GEMM or softmax kernel in the step."""
        from .retrain_head import RetrainHeadParams
        import ctypes as C
        native, lib, st = self.native, self.lib, torch.cuda.current_stream().cuda_stream
        W = self.param("final_training_ops/weights/final_weights")
        b = self.param("final_training_ops/biases/final_biases")
        B = self.B
        if not hasattr(self, "_head_dz"):
            self._head_dz = torch.empty(B, self.C, dtype=torch.float32, device=self.device)
            # two (loss sum, correct count) slots used in turn: a step's last head kernel clears the other one, and
            # metrics() divides on the host, so the step holds no memset or element-wise launch
            self._head_metrics = torch.zeros(2, 2, dtype=torch.float32, device=self.device)
            self._head_par = 0
            self._head_labels = torch.empty(B, dtype=torch.int32, device=self.device)
        self._head_labels.copy_(labels)
        cur, nxt = self._head_metrics[self._head_par], self._head_metrics[self._head_par ^ 1]
        self._head_par ^= 1
        p = RetrainHeadParams()
        p.cache, p.labels_all, p.rows = self.feat.data_ptr(), self._head_labels.data_ptr(), None
        p.W, p.b, p.B, p.C, p.grad_scale = W.data_ptr(), b.data_ptr(), B, self.C, 1.0 / B
        p.double_softmax, p.training = int(self.double_softmax), 1
        p.probs, p.pred, p.dz, p.metrics = None, None, self._head_dz.data_ptr(), cur.data_ptr()
        native.check(lib.dtf_retrain_head_fwd(C.byref(p), st), "head fwd")
        native.check(lib.dtf_retrain_head_bwd(self.feat.data_ptr(), None, self._head_dz.data_ptr(), B, self.C, None, None,
                                              self.grad("final_training_ops/weights/final_weights").data_ptr(),
                                              self.grad("final_training_ops/biases/final_biases").data_ptr(), 0.0, st), "head bwd")
        native.check(lib.dtf_retrain_head_dx(self._head_dz.data_ptr(), W.data_ptr(), self.dfeat.data_ptr(), B, self.C,
                                             nxt.data_ptr(), st), "head dx")
        # (loss sum, correct count) stay on the device until the next step clears that slot; metrics() reads them
        self._last_metrics = cur

    def _dw_splits(self, K: int, Cout: int, npos: int) -> int:
        blocks = ((K + 127) // 128) * ((Cout + 63) // 64)
        want = max(1, 1536 // blocks)
        return max(1, min(want, npos // 512 if npos >= 512 else 1))

    @staticmethod
    def _dw2_splits(s, npos: int, variant: int = 0) -> int:
        """Split-K factor of the filter-gradient kernel.  Every split adds its whole tile to dW with fp32 atomics, and
        those — not the MFMAs — bound the kernel (measured: 1024 workgroups x 16K atomics = 80 us per conv), so use the
        fewest splits that still put one resident workgroup on every CU (96 KB of LDS: one 128-tile workgroup per CU,
        three 64-tile ones)."""
        bt = 128 if (s.cin > 64 and s.cout > 64 and not variant) else 64
        tiles = ((s.kh * s.kw * s.cin + bt - 1) // bt) * ((s.cout + bt - 1) // bt)  # (filter rows are flattened over taps)
        target = int(os.environ.get("DTF_DW_BLOCKS", "0")) or (256 if bt == 128 else 768)
        # round DOWN: one workgroup more than fits at once costs a whole extra round (measured: 280 workgroups of a
        # 256-slot grid took 2 x 37 us)
        return max(1, min(target // tiles, npos // 256 if npos >= 256 else 1))

    def backward(self, on_conv_done=None) -> None:
        """Runs the backward program; ``on_conv_done(name)`` is called after the kernels that finalise conv ``name``'s
        arena segment have been enqueued (the DP trainer launches bucket all-reduces from it)."""
        g, B, lib, chk = self.graph, self.B, self.lib, self.native.check
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        x = self._x
        par = self._bwd_parity
        self._bwd_parity ^= 1
        # A layer's filter gradient feeds nothing but the optimizer, and it is the longest launch of the layer: it runs on
        # a second stream next to the chain (batch-norm backward -> data gradient -> next layer) that the rest of the
        # backward pass waits on, filling the CUs the chain's partial rounds leave idle.  One event per layer main -> side
        # (the gradient w.r.t. the conv output is final), one side -> main at the end.  (Measured at batch 128: 21.1 -> 19.9 ms
        # per step.  The chain itself spread over 2-4 streams by branch, with the read-modify-write order of the shared
        # input gradients kept by events, was slower: 18.9 -> 19.6-20.3 ms.)
        dw = None
        if self.dw_side:
            if self._dw_stream is None:
                self._dw_stream = torch.cuda.Stream(device=self.device)
                self._dw_ev = {i: torch.cuda.Event() for i, o in enumerate(g.ops) if o.kind == "conv"}
                self._dw_done = torch.cuda.Event()
            dw = self._dw_stream
        for i in range(len(g.ops) - 1, -1, -1):
            op = g.ops[i]
            src = x if op.src == "input" else self.act[op.src]
            _, H, W, Cs = src.shape
            if op.kind == "gap":
                chk(lib.dtf_global_avgpool_bwd(self.dfeat.data_ptr(), self.dact[op.src].data_ptr(), B, H * W, Cs, st), "gap_bwd")
                continue
            dst, ddst = self.act[op.dst], self.dact[op.dst]
            _, Ho, Wo, Cd = dst.shape
            if op.kind == "conv":
                s = op.conv
                z = self.z[s.name]
                M = B * Ho * Wo
                dy_p = ddst.data_ptr() + 2 * op.dst_off
                cur, prev = self._stats(s.name, 1 + par), self._stats(s.name, 2 - par)
                chk(lib.dtf_bn_relu_bwd_reduce(dy_p, Cd, z.data_ptr(), self._bn_ptr(s.name), cur, M, s.cout, st),
                    "bn_bwd_reduce")
                chk(lib.dtf_bn_relu_bwd_apply(dy_p, Cd, z.data_ptr(), self._bn_ptr(s.name), cur, prev,
                                              self.grad(s.name + "/batchnorm/gamma").data_ptr(),
                                              self.grad(s.name + "/batchnorm/beta").data_ptr(), M, s.cout, st), "bn_bwd_apply")
                if dw is not None:
                    self._dw_ev[i].record(main)
                    dw.wait_event(self._dw_ev[i])
                self._conv_dw(op, src, self.core[s.name]["dw"], st if dw is None else dw.cuda_stream)
                if op.src != "input":
                    self._conv_bwd_data(op, src, self.core[s.name]["bwd"], int(self.accumulate[i]), st)
                if on_conv_done is not None:
                    # (the hook orders its work after the CURRENT stream: make that the stream of the layer's last writer)
                    with torch.cuda.stream(dw if dw is not None else main):
                        on_conv_done(s.name)
            elif op.kind == "maxpool":
                chk(lib.dtf_maxpool3x3_bwd(ddst.data_ptr() + 2 * op.dst_off, Cd, self.argmax[i].data_ptr(),
                                           self.dact[op.src].data_ptr(), Cs, B, H, W, op.channels, op.stride,
                                           int(self.accumulate[i]), st), "maxpool_bwd")
            elif op.kind == "avgpool":
                chk(lib.dtf_avgpool3x3_same_bwd(ddst.data_ptr(), Cd, self.dact[op.src].data_ptr(), Cs, B, H, W, op.channels,
                                                int(self.accumulate[i]), st), "avgpool_bwd")
        if dw is not None:
            self._dw_done.record(dw)
            main.wait_event(self._dw_done)

    def apply(self) -> None:
        from ..ops import hip_ops
        scale = 1.0 / self.world
        if self.optimizer == "adam":
            hip_ops.adam_tf_(self.P, self.G, self.M, self.V, self.step_t, self.ticket, lr=self.lr, grad_scale=scale,
                             zero_from=0)
        else:
            hip_ops.sgd_(self.P, self.G, self.step_t, self.ticket, lr=self.lr, grad_scale=scale, zero_from=0)

    def fwd_bwd(self, x: torch.Tensor, labels: torch.Tensor, on_conv_done=None) -> None:
        self.forward(x, training=True)
        self.head_forward_backward(labels)
        self.backward(on_conv_done)

    def train_step(self, x: torch.Tensor, labels: torch.Tensor) -> None:
        """One full step: forward, loss, backward, optimizer (which also re-zeroes the gradient arena)."""
        self.fwd_bwd(x, labels)
        self.apply()

    def metrics(self) -> Tuple[float, float]:
        loss_sum, correct = self._last_metrics.tolist()
        return loss_sum / self.B, correct / self.B

    def evaluate_batch(self, x: torch.Tensor, labels: torch.Tensor, out: torch.Tensor) -> None:
        """Forward only (batch statistics, moving statistics untouched): the batch's loss sum and correct count are ADDED
        to ``out`` (fp32 [2] on the device; no host sync).  Call it an even number of times between two training steps:
        the forward alternates between two sets of statistic accumulators, and a captured step graph assumes the set it
        was captured with."""
        from .retrain_head import RetrainHeadParams
        import ctypes as C
        self.forward(x, training=False)
        if not hasattr(self, "_eval_labels"):
            self._eval_labels = torch.empty(self.B, dtype=torch.int32, device=self.device)
        self._eval_labels.copy_(labels)
        p = RetrainHeadParams()
        p.cache, p.labels_all, p.rows = self.feat.data_ptr(), self._eval_labels.data_ptr(), None
        p.W = self.param("final_training_ops/weights/final_weights").data_ptr()
        p.b = self.param("final_training_ops/biases/final_biases").data_ptr()
        p.B, p.C, p.grad_scale, p.double_softmax, p.training = self.B, self.C, 0.0, int(self.double_softmax), 0
        p.probs, p.pred, p.dz, p.metrics = None, None, None, out.data_ptr()
        self.native.check(self.lib.dtf_retrain_head_fwd(C.byref(p), torch.cuda.current_stream().cuda_stream), "head eval")

    def state_tensors(self) -> Dict[str, torch.Tensor]:
        """Everything a checkpoint of the fine-tune holds: parameters, optimizer slots, moving statistics, step counter."""
        out = {"P": self.P, "moving_mean": self.moving_mean, "moving_var": self.moving_var, "step": self.step_t}
        if self.optimizer == "adam":
            out.update(M=self.M, V=self.V)
        return out

    def flops_per_step(self) -> float:
        from .inception_v3 import graph_flops
        return 3.0 * graph_flops(self.graph) * self.B


# ------------------------------------------------------------------------------------------------ PyTorch oracle
def torch_forward_train(graph, params: Dict[str, torch.Tensor], x_nchw: torch.Tensor, *, bf16_sim: bool = False,
                        channels_last: bool = False, return_buffers: bool = False):
    """Training-mode (batch statistics) forward of the op program with differentiable torch ops.
    ``params[name + '/conv2d_params' | '/batchnorm/gamma' | '/batchnorm/beta']``; returns the [B,2048] features."""
    r = (lambda t: t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()) if bf16_sim else (lambda t: t)
    bufs: Dict[str, List[Tuple[int, torch.Tensor]]] = {}
    whole: Dict[str, torch.Tensor] = {"input": x_nchw}

    def get(name):
        if name not in whole:
            parts = sorted(bufs[name], key=lambda t: t[0])
            whole[name] = torch.cat([p for _o, p in parts], dim=1)
        return whole[name]

    feat = None
    for op in graph.ops:
        src = get(op.src)
        if op.kind == "conv":
            s = op.conv
            w = params[s.name + "/conv2d_params"]
            w = r(w[:, :, :src.shape[1], :]).permute(3, 2, 0, 1)
            if channels_last:
                w = w.contiguous(memory_format=torch.channels_last)
            z = r(F.conv2d(src, w.to(src.dtype), None, stride=s.stride, padding=s.pad))
            zf = z.float()
            mean = zf.mean((0, 2, 3), keepdim=True)
            var = zf.var((0, 2, 3), unbiased=False, keepdim=True)
            gmm = params[s.name + "/batchnorm/gamma"].view(1, -1, 1, 1)
            bta = params[s.name + "/batchnorm/beta"].view(1, -1, 1, 1)
            y = r(F.relu((zf - mean) * torch.rsqrt(var + BN_EPS) * gmm + bta)).to(src.dtype)
        elif op.kind == "maxpool":
            y = F.max_pool2d(src, 3, op.stride)
        elif op.kind == "avgpool":
            y = r(F.avg_pool2d(src, 3, 1, 1, count_include_pad=False))
        elif op.kind == "gap":
            feat = src.float().mean((2, 3))
            continue
        bufs.setdefault(op.dst, []).append((op.dst_off, y))
    if return_buffers:
        return feat, {name: get(name) for name in bufs}
    return feat


def torch_train_step(graph, params: Dict[str, torch.Tensor], x_nchw: torch.Tensor, labels: torch.Tensor, *,
                     double_softmax: bool = False, bf16_sim: bool = False):
    """Oracle: (loss, {name: grad}) of one step with autograd."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    feat = torch_forward_train(graph, leaf, x_nchw, bf16_sim=bf16_sim)
    logits = feat @ leaf["final_training_ops/weights/final_weights"] + leaf["final_training_ops/biases/final_biases"]
    if double_softmax:
        loss = F.cross_entropy(torch.softmax(logits, 1), labels.long())
    else:
        loss = F.cross_entropy(logits, labels.long())
    grads = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
    return loss.detach(), dict(zip(leaf.keys(), grads)), feat.detach()


class EagerInceptionTrainer:
    """Same interface as ``InceptionTrainer`` on plain torch autograd (any device, fp32): the CPU path of the
    fine-tune application and of the multi-process gloo tests."""

    def __init__(self, num_classes: int, batch_size: int, device="cpu", *, lr: float = 0.01, optimizer: str = "sgd",
                 seed: int = 0, init: Optional[InceptionV3] = None, double_softmax: bool = False, head_std: float = 0.001,
                 **_unused):
        if optimizer != "sgd":
            raise ValueError("the eager fine-tune path implements SGD only")
        self.device = torch.device(device)
        self.B, self.C, self.lr, self.double_softmax, self.world = batch_size, num_classes, lr, double_softmax, 1
        self.graph = build_graph()
        self.layout = ArenaLayout(self.graph, num_classes)
        self.P = torch.zeros(self.layout.size, dtype=torch.float32, device=self.device)
        self.G = torch.zeros_like(self.P)
        self.step_t = torch.zeros(1, dtype=torch.int32, device=self.device)
        src = init if init is not None else InceptionV3(device="cpu", seed=seed)
        gen = torch.Generator(device="cpu").manual_seed(seed + 17)
        hw = torch.empty(BOTTLENECK_TENSOR_SIZE, num_classes)
        torch.nn.init.trunc_normal_(hw, std=head_std, a=-2 * head_std, b=2 * head_std, generator=gen)
        self.param("final_training_ops/weights/final_weights").copy_(hw)
        for name, s in self.graph.convs.items():
            p = src.params[name]
            w = p["weights"]
            if w.shape[2] != s.cin:
                w = torch.cat([w, torch.zeros(s.kh, s.kw, s.cin - w.shape[2], s.cout)], dim=2)
            self.param(name + "/conv2d_params").copy_(w)
            self.param(name + "/batchnorm/gamma").copy_(p["gamma"])
            self.param(name + "/batchnorm/beta").copy_(p["beta"])
        self.last_loss = torch.zeros(())
        self.last_correct = torch.zeros(())

    def param(self, name: str) -> torch.Tensor:
        return self.layout.view(self.P, name)

    def grad(self, name: str) -> torch.Tensor:
        return self.layout.view(self.G, name)

    def fwd_bwd(self, x: torch.Tensor, labels: torch.Tensor, on_conv_done=None) -> None:
        """``x``: NHWC ``[B,299,299,>=3]`` (the native layout; extra pad channels ignored)."""
        leaf = {k: self.layout.view(self.P, k).detach().clone().requires_grad_(True) for k in self.layout.entries}
        xin = x[..., :3].float().permute(0, 3, 1, 2).contiguous()
        feat = torch_forward_train(self.graph, leaf, xin)
        logits = feat @ leaf["final_training_ops/weights/final_weights"] + leaf["final_training_ops/biases/final_biases"]
        loss = F.cross_entropy(torch.softmax(logits, 1) if self.double_softmax else logits, labels.long())
        grads = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
        for (k, _v), g in zip(leaf.items(), grads):
            if g is not None:
                self.grad(k).copy_(g)
        self.last_loss = loss.detach()
        self.last_correct = (logits.argmax(1) == labels.long()).float().sum()
        if on_conv_done is not None:
            for name in self.layout.conv_order:
                on_conv_done(name)

    def apply(self) -> None:
        self.P.add_(self.G, alpha=-self.lr / self.world)
        self.G.zero_()
        self.step_t += 1

    def train_step(self, x, labels) -> None:
        self.fwd_bwd(x, labels)
        self.apply()

    def metrics(self) -> Tuple[float, float]:
        return float(self.last_loss), float(self.last_correct) / self.B

    @torch.no_grad()
    def evaluate_batch(self, x: torch.Tensor, labels: torch.Tensor, out: torch.Tensor) -> None:
        params = {k: self.layout.view(self.P, k) for k in self.layout.entries}
        feat = torch_forward_train(self.graph, params, x[..., :3].float().permute(0, 3, 1, 2).contiguous())
        logits = feat @ params["final_training_ops/weights/final_weights"] + params["final_training_ops/biases/final_biases"]
        z = torch.softmax(logits, 1) if self.double_softmax else logits
        out[0] += F.cross_entropy(z, labels.long(), reduction="sum")
        out[1] += (logits.argmax(1) == labels.long()).float().sum()

    def state_tensors(self) -> Dict[str, torch.Tensor]:
        return {"P": self.P, "step": self.step_t}

    def export_to(self, model: InceptionV3) -> None:
        """Weights into an inference network.  This CPU plumbing path keeps no moving statistics (training-mode batch
        norm only): the exported batch-norm constants are identity statistics."""
        layers = {}
        for name, s in self.graph.convs.items():
            w = self.param(name + "/conv2d_params").detach().cpu()
            layers[name] = {"weights": w[:, :, :model.params[name]["weights"].shape[2], :].contiguous().numpy(),
                            "gamma": self.param(name + "/batchnorm/gamma").cpu().numpy(),
                            "beta": self.param(name + "/batchnorm/beta").cpu().numpy()}
        model.load_tf_weights(layers)


def calibrate_batchnorm(model: InceptionV3, x: torch.Tensor) -> None:
    """Data-dependent initialisation of a *random-init* backbone: set every conv's BatchNorm moving statistics to the
    batch statistics of one calibration batch (bf16 NHWC ``[B,299,299,8]``).  Without pretrained weights the folded
    identity statistics let activation scale drift over 47 layers and the 2048-d features stop separating images;
    with calibrated statistics random convolutional features are informative enough for the retrain demo."""
    tr = InceptionTrainer(2, x.shape[0], x.device, init=model, bn_momentum=0.0, conv_core="v1")
    tr.forward(x.contiguous(), training=True)  # momentum 0: moving statistics := batch statistics
    torch.cuda.synchronize()
    tr.export_to(model)


def make_inception_trainer(kind: str, *args, **kw):
    if kind == "native":
        return InceptionTrainer(*args, **kw)
    if kind == "eager":
        return EagerInceptionTrainer(*args, **kw)
    raise ValueError("unknown fine-tune engine %r" % kind)
