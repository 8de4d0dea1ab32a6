"""Inception-v3 (2015-12-05 ``classify_image_graph_def.pb`` topology) — forward to the 2048-d
``pool_3/_reshape`` bottleneck (components C28, C31).

The reference imports the frozen TensorFlow graph and runs it one image per ``sess.run``
(``retrain1/retrain.py:66-74,228-231``).  Here the network is stated natively as a small op
program over named NHWC bf16 buffers and executed in *batches* on the GPU:

* every conv = implicit-GEMM MFMA kernel with the inference BatchNorm folded into the packed
  bf16 weights + fp32 bias and ReLU in the epilogue (``dtf_conv_bias_relu``);
* branch outputs are written straight into their channel slice of the block's output buffer —
  there is no concat kernel and no concat copy;
* pools are vectorised NHWC kernels (``csrc/hip/inception_ops.hip``); the final 8x8 average pool
  emits the fp32 bottleneck rows that go into the HBM-resident cache;
* the whole forward for a fixed batch size is captured once into a hipGraph (≈110 launches).

Weights are random-init (He-normal convs, identity BatchNorm statistics) because the pretrained
``.pb`` cannot be downloaded here (``maybe_download_and_extract``, ``retrain1/retrain.py:47-62``);
``load_tf_weights`` accepts a ``{layer: {weights, beta, moving_mean, moving_variance}}`` dict for
when the artefact is available.  ``reference_forward`` is the PyTorch fp32 oracle of the same graph.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BOTTLENECK_TENSOR_SIZE = 2048      # retrain1/retrain.py:31
MODEL_INPUT_WIDTH = 299            # :32
MODEL_INPUT_HEIGHT = 299           # :33
MODEL_INPUT_DEPTH = 3              # :34
INPUT_PAD_CHANNELS = 8             # NHWC input is zero-padded 3 -> 8 channels (16-byte gathers)
BN_EPS = 1e-3


@dataclass
class ConvSpec:
    name: str
    cin: int
    cout: int
    kh: int
    kw: int
    stride: int
    pad: Tuple[int, int]


@dataclass
class Op:
    kind: str            # conv | maxpool | avgpool | gap
    src: str
    dst: str
    src_off: int = 0     # channel offset inside src
    dst_off: int = 0     # channel offset inside dst
    conv: Optional[ConvSpec] = None
    channels: int = 0    # for pools
    stride: int = 1


class _Builder:
    def __init__(self):
        self.ops: List[Op] = []
        self.shapes: "OrderedDict[str, Tuple[int, int, int]]" = OrderedDict()  # buffer -> (H, W, C)
        self.convs: "OrderedDict[str, ConvSpec]" = OrderedDict()

    def buffer(self, name: str, h: int, w: int, c: int) -> str:
        self.shapes[name] = (h, w, c)
        return name

    def conv(self, name, src, cout, k, stride=1, padding="SAME", dst=None, dst_off=0, src_off=0, cin=None):
        kh, kw = (k, k) if isinstance(k, int) else k
        h, w, c_total = self.shapes[src]
        cin = c_total if cin is None else cin
        ph, pw = ((kh - 1) // 2, (kw - 1) // 2) if padding == "SAME" else (0, 0)
        ho, wo = (h + 2 * ph - kh) // stride + 1, (w + 2 * pw - kw) // stride + 1
        if dst is None:
            dst = self.buffer(name + ":out", ho, wo, cout)
        assert self.shapes[dst][:2] == (ho, wo), (name, self.shapes[dst], ho, wo)
        spec = ConvSpec(name, cin, cout, kh, kw, stride, (ph, pw))
        self.convs[name] = spec
        self.ops.append(Op("conv", src, dst, src_off, dst_off, spec))
        return dst

    def maxpool(self, name, src, stride=2, dst=None, dst_off=0):
        h, w, c = self.shapes[src]
        ho, wo = (h - 3) // stride + 1, (w - 3) // stride + 1
        if dst is None:
            dst = self.buffer(name + ":out", ho, wo, c)
        self.ops.append(Op("maxpool", src, dst, 0, dst_off, None, c, stride))
        return dst

    def avgpool(self, name, src):
        h, w, c = self.shapes[src]
        dst = self.buffer(name + ":out", h, w, c)
        self.ops.append(Op("avgpool", src, dst, 0, 0, None, c, 1))
        return dst


def build_graph() -> _Builder:
    g = _Builder()
    x = g.buffer("input", 299, 299, INPUT_PAD_CHANNELS)
    x = g.conv("conv", x, 32, 3, 2, "VALID")          # 149
    x = g.conv("conv_1", x, 32, 3, 1, "VALID")        # 147
    x = g.conv("conv_2", x, 64, 3, 1, "SAME")         # 147
    x = g.maxpool("pool", x, 2)                       # 73
    x = g.conv("conv_3", x, 80, 1, 1, "VALID")        # 73
    x = g.conv("conv_4", x, 192, 3, 1, "VALID")       # 71
    x = g.maxpool("pool_1", x, 2)                     # 35

    def block_a(name, src, pool_proj):
        h, w, _ = g.shapes[src]
        out = g.buffer(name, h, w, 64 + 64 + 96 + pool_proj)
        g.conv(name + "/conv", src, 64, 1, dst=out, dst_off=0)
        t = g.conv(name + "/tower/conv", src, 48, 1)
        g.conv(name + "/tower/conv_1", t, 64, 5, dst=out, dst_off=64)
        t = g.conv(name + "/tower_1/conv", src, 64, 1)
        t = g.conv(name + "/tower_1/conv_1", t, 96, 3)
        g.conv(name + "/tower_1/conv_2", t, 96, 3, dst=out, dst_off=128)
        t = g.avgpool(name + "/tower_2/pool", src)
        g.conv(name + "/tower_2/conv", t, pool_proj, 1, dst=out, dst_off=224)
        return out

    x = block_a("mixed", x, 32)       # 256
    x = block_a("mixed_1", x, 64)     # 288
    x = block_a("mixed_2", x, 64)     # 288

    # mixed_3: 35 -> 17, 768 channels
    h, w, c = g.shapes[x]
    out = g.buffer("mixed_3", 17, 17, 384 + 96 + c)
    g.conv("mixed_3/conv", x, 384, 3, 2, "VALID", dst=out, dst_off=0)
    t = g.conv("mixed_3/tower/conv", x, 64, 1)
    t = g.conv("mixed_3/tower/conv_1", t, 96, 3)
    g.conv("mixed_3/tower/conv_2", t, 96, 3, 2, "VALID", dst=out, dst_off=384)
    g.maxpool("mixed_3/pool", x, 2, dst=out, dst_off=480)
    x = out

    def block_c(name, src, c7):
        out = g.buffer(name, 17, 17, 768)
        g.conv(name + "/conv", src, 192, 1, dst=out, dst_off=0)
        t = g.conv(name + "/tower/conv", src, c7, 1)
        t = g.conv(name + "/tower/conv_1", t, c7, (1, 7))
        g.conv(name + "/tower/conv_2", t, 192, (7, 1), dst=out, dst_off=192)
        t = g.conv(name + "/tower_1/conv", src, c7, 1)
        t = g.conv(name + "/tower_1/conv_1", t, c7, (7, 1))
        t = g.conv(name + "/tower_1/conv_2", t, c7, (1, 7))
        t = g.conv(name + "/tower_1/conv_3", t, c7, (7, 1))
        g.conv(name + "/tower_1/conv_4", t, 192, (1, 7), dst=out, dst_off=384)
        t = g.avgpool(name + "/tower_2/pool", src)
        g.conv(name + "/tower_2/conv", t, 192, 1, dst=out, dst_off=576)
        return out

    x = block_c("mixed_4", x, 128)
    x = block_c("mixed_5", x, 160)
    x = block_c("mixed_6", x, 160)
    x = block_c("mixed_7", x, 192)

    # mixed_8: 17 -> 8, 1280 channels
    out = g.buffer("mixed_8", 8, 8, 320 + 192 + 768)
    t = g.conv("mixed_8/tower/conv", x, 192, 1)
    g.conv("mixed_8/tower/conv_1", t, 320, 3, 2, "VALID", dst=out, dst_off=0)
    t = g.conv("mixed_8/tower_1/conv", x, 192, 1)
    t = g.conv("mixed_8/tower_1/conv_1", t, 192, (1, 7))
    t = g.conv("mixed_8/tower_1/conv_2", t, 192, (7, 1))
    g.conv("mixed_8/tower_1/conv_3", t, 192, 3, 2, "VALID", dst=out, dst_off=320)
    g.maxpool("mixed_8/pool", x, 2, dst=out, dst_off=512)
    x = out

    def block_e(name, src):
        out = g.buffer(name, 8, 8, 2048)
        g.conv(name + "/conv", src, 320, 1, dst=out, dst_off=0)
        t = g.conv(name + "/tower/conv", src, 384, 1)
        g.conv(name + "/tower/mixed/conv", t, 384, (1, 3), dst=out, dst_off=320)
        g.conv(name + "/tower/mixed/conv_1", t, 384, (3, 1), dst=out, dst_off=704)
        t = g.conv(name + "/tower_1/conv", src, 448, 1)
        t = g.conv(name + "/tower_1/conv_1", t, 384, 3)
        g.conv(name + "/tower_1/mixed/conv", t, 384, (1, 3), dst=out, dst_off=1088)
        g.conv(name + "/tower_1/mixed/conv_1", t, 384, (3, 1), dst=out, dst_off=1472)
        t = g.avgpool(name + "/tower_2/pool", src)
        g.conv(name + "/tower_2/conv", t, 192, 1, dst=out, dst_off=1856)
        return out

    x = block_e("mixed_9", x)
    x = block_e("mixed_10", x)
    g.ops.append(Op("gap", x, "pool_3", 0, 0, None, 2048, 1))
    return g


def graph_flops(g: _Builder) -> float:
    """Forward FLOPs per image (2 x MACs)."""
    total = 0.0
    for op in g.ops:
        if op.kind == "conv":
            ho, wo, _ = g.shapes[op.dst]
            s = op.conv
            cin = 3 if s.name == "conv" else s.cin
            total += 2.0 * ho * wo * s.cout * s.kh * s.kw * cin
    return total


class StreamPlan:
    """The op program over ``K`` HIP streams: static list scheduling (an op goes where it finishes first by the given cost
    estimates; a dependency that crosses streams costs ``edge_us``, an event wait between two hardware queues was measured at
    5-10 us) plus the events that carry the dependencies.  The branches of a mixed block are independent until the concat and
    no single launch of the 17x17 / 8x8 layers fills 256 CUs for its whole duration: a second and third stream fill the
    partial rounds.  Dependencies are buffer-level: an op waits for every earlier op that wrote its source buffer.

    ``begin(main)`` / ``stream_for(i)`` (waits issued, returns the stream op ``i`` runs on) / ``done(i)`` / ``end()`` (every
    side stream joined back into ``main``) — fork / join in the form stream capture wants, so the same code runs eagerly and
    inside a hipGraph capture.  An edge between two SIDE streams is relayed through ``main``: ending the capture of a graph
    that holds such an edge crashes inside the HIP runtime (ROCm 7.0.2), and the relay costs nothing here because those
    edges are the concat points, where everything waits for everything."""

    def __init__(self, ops, costs_us, K: int, device, edge_us: float = 6.0):
        edge_us = float(os.environ.get("DTF_STREAM_EDGE_US", edge_us))
        n = len(ops)
        writers: Dict[str, List[int]] = {}
        avail, finish, stream = [0.0] * K, [0.0] * n, [0] * n
        waits: List[List[int]] = [[] for _ in range(n)]
        for i, op in enumerate(ops):
            deps = writers.get(op.src, [])
            best = None
            for k in range(K):
                ready = max([avail[k]] + [finish[d] + (edge_us if stream[d] != k else 0.0) for d in deps])
                if best is None or ready + costs_us[i] < best[0] - 1e-9:
                    best = (ready + costs_us[i], k)
            finish[i], stream[i] = best
            avail[stream[i]] = finish[i]
            latest: Dict[int, int] = {}
            for d in deps:  # (stream order: the latest writer on each other stream covers the earlier ones)
                if stream[d] != stream[i]:
                    latest[stream[d]] = max(latest.get(stream[d], -1), d)
            waits[i] = sorted(latest.values())
            writers.setdefault(op.dst, []).append(i)
        self.stream_of, self.waits = stream, waits
        self.waited = {d for w in waits for d in w}
        self.serial_us, self.planned_us = float(sum(costs_us)), float(max(finish)) if n else 0.0
        self.K, self.device = K, device
        self.used = sorted({k for k in stream if k != 0})
        self.side = None   # streams and events are made at the first begin(): the plan itself needs no GPU

    def begin(self, main) -> None:
        if self.side is None:
            self.side = [torch.cuda.Stream(device=self.device) for _ in range(self.K - 1)]
            self.ev = {d: torch.cuda.Event() for d in self.waited}
            self.relay = {d: torch.cuda.Event() for d in self.waited if self.stream_of[d] != 0}
            self.join = [torch.cuda.Event() for _ in self.side]
        self.main, self.streams, self.relayed = main, [main] + self.side, set()

    def stream_for(self, i: int):
        k = self.stream_of[i]
        s = self.streams[k]
        for d in self.waits[i]:
            if k != 0 and self.stream_of[d] != 0:
                if d not in self.relayed:
                    self.main.wait_event(self.ev[d])
                    self.relay[d].record(self.main)
                    self.relayed.add(d)
                s.wait_event(self.relay[d])
            else:
                s.wait_event(self.ev[d])
        return s

    def done(self, i: int) -> None:
        if i in self.waited:
            self.ev[i].record(self.streams[self.stream_of[i]])

    def end(self) -> None:
        for k in self.used:
            self.join[k - 1].record(self.streams[k])
            self.main.wait_event(self.join[k - 1])


class InceptionV3:
    """Weights + native batched executor."""

    def __init__(self, device="cuda", seed: Optional[int] = 0, conv_core: str = "auto"):
        self.graph = build_graph()
        self.conv_core = conv_core       # "v1" | "v2" | "auto" (time both GEMM cores per layer and batch size once)
        self._core: Dict[int, Dict[str, int]] = {}
        self._core_us: Dict[int, Dict[str, float]] = {}   # measured duration of the chosen core per layer (stream planner)
        self._plans: Dict[int, "StreamPlan"] = {}
        self.device = torch.device(device)
        self.params: "OrderedDict[str, Dict[str, torch.Tensor]]" = OrderedDict()
        gen = torch.Generator(device="cpu")
        gen.manual_seed(0 if seed is None else seed)
        for name, s in self.graph.convs.items():
            cin_real = MODEL_INPUT_DEPTH if name == "conv" else s.cin
            fan_in = s.kh * s.kw * cin_real
            w = torch.empty(s.kh, s.kw, cin_real, s.cout)
            torch.nn.init.trunc_normal_(w, std=math.sqrt(2.0 / fan_in), a=-2 * math.sqrt(2.0 / fan_in),
                                        b=2 * math.sqrt(2.0 / fan_in), generator=gen)
            self.params[name] = {"weights": w, "beta": torch.zeros(s.cout), "moving_mean": torch.zeros(s.cout),
                                 "moving_variance": torch.ones(s.cout), "gamma": torch.ones(s.cout)}
        self._packed: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._buffers: Dict[int, Dict[str, torch.Tensor]] = {}
        self._graphs: Dict[int, object] = {}
        self._static_in: Dict[int, torch.Tensor] = {}
        self._static_out: Dict[int, torch.Tensor] = {}

    # ------------------------------------------------------------------ weights
    @property
    def num_params(self) -> int:
        return sum(int(p["weights"].numel()) + 2 * int(p["beta"].numel()) for p in self.params.values())

    def load_tf_weights(self, tensors: Dict[str, Dict[str, np.ndarray]]) -> None:
        for name, d in tensors.items():
            if name not in self.params:
                raise KeyError("unknown Inception layer %r" % name)
            for k, v in d.items():
                t = torch.as_tensor(np.asarray(v), dtype=torch.float32)
                if t.shape != self.params[name][k].shape:
                    raise ValueError("%s/%s: shape %s != %s" % (name, k, tuple(t.shape), tuple(self.params[name][k].shape)))
                self.params[name][k] = t
        self._packed.clear()

    def export_folded(self) -> "OrderedDict[str, torch.Tensor]":
        """The inference state of the backbone: per conv layer the BN-folded HWIO filter as bf16 (exactly the bits the HIP
        conv consumes) and the fp32 bias, ~44 MB — what a self-contained retrained bundle stores
        (``retrain1/retrain.py:470-473`` freezes backbone + head into one file)."""
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for name in self.graph.convs:
            w, b = self.folded(name)
            out["backbone/%s/folded_weights" % name] = w.to(torch.bfloat16).contiguous()
            out["backbone/%s/folded_bias" % name] = b.to(torch.float32).contiguous()
        return out

    def load_folded(self, tensors) -> None:
        """Inverse of ``export_folded`` (bf16 entries may arrive as uint16 bit patterns): identity batch norm around the
        stored folded filter, so ``packed()`` reproduces the exporting model's operands bit for bit."""
        layers = {}
        for name, p in self.params.items():
            w = tensors["backbone/%s/folded_weights" % name]
            w = torch.as_tensor(np.asarray(w))
            if w.dtype in (torch.uint16, torch.int16):
                w = w.view(torch.bfloat16)
            b = torch.as_tensor(np.asarray(tensors["backbone/%s/folded_bias" % name]), dtype=torch.float32)
            c = b.shape[0]
            layers[name] = {"weights": w.float().numpy(), "beta": b.numpy(), "gamma": np.ones(c, np.float32),
                            "moving_mean": np.zeros(c, np.float32),
                            "moving_variance": np.full(c, 1.0 - BN_EPS, np.float32)}  # gamma / sqrt(var + eps) == 1
        self.load_tf_weights(layers)

    def folded(self, name: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """(HWIO fp32 weights with BN folded, fp32 bias)."""
        p = self.params[name]
        scale = p["gamma"] / torch.sqrt(p["moving_variance"] + BN_EPS)
        return p["weights"] * scale.view(1, 1, 1, -1), p["beta"] - p["moving_mean"] * scale

    def packed(self, name: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """bf16 [Cout][KH*KW*Cin_padded] + fp32 bias on the device (what the HIP conv consumes)."""
        if name not in self._packed:
            s = self.graph.convs[name]
            w, b = self.folded(name)
            if w.shape[2] != s.cin:  # first conv: pad 3 -> 8 input channels with zeros
                w = torch.cat([w, torch.zeros(s.kh, s.kw, s.cin - w.shape[2], s.cout)], dim=2)
            wp = w.permute(3, 0, 1, 2).reshape(s.cout, s.kh * s.kw * s.cin).contiguous()
            self._packed[name] = (wp.to(self.device, torch.bfloat16).contiguous(), b.to(self.device, torch.float32).contiguous())
        return self._packed[name]

    # ------------------------------------------------------------------ native execution
    def _alloc(self, B: int) -> Dict[str, torch.Tensor]:
        if B not in self._buffers:
            bufs = {}
            for name, (h, w, c) in self.graph.shapes.items():
                if name == "input":
                    continue
                bufs[name] = torch.zeros(B, h, w, c, dtype=torch.bfloat16, device=self.device)
            bufs["pool_3"] = torch.zeros(B, BOTTLENECK_TENSOR_SIZE, dtype=torch.float32, device=self.device)
            self._buffers[B] = bufs
        return self._buffers[B]

    def _run_ops(self, x: torch.Tensor, bufs: Dict[str, torch.Tensor]) -> torch.Tensor:
        from .. import native
        from ..ops import hip_ops
        lib = native.hip()
        main = torch.cuda.current_stream()
        st = main.cuda_stream
        B = x.shape[0]
        plan = self._stream_plan(B, bufs)
        if plan is not None:
            plan.begin(main)
        for i, op in enumerate(self.graph.ops):
            src = x if op.src == "input" else bufs[op.src]
            cur = main if plan is None else plan.stream_for(i)
            st = cur.cuda_stream
            if op.kind == "conv":
                s = op.conv
                wp, b = self.packed(s.name)
                with torch.cuda.stream(cur):  # (the op wrapper launches on the current stream)
                    hip_ops.conv_bias_relu(src, wp, b, Cin=s.cin, KH=s.kh, KW=s.kw, stride=s.stride, pad=s.pad, relu=True,
                                           out=bufs[op.dst], out_channel_offset=op.dst_off, in_channel_offset=op.src_off,
                                           core=self._core.get(B, {}).get(s.name, 1))
            elif op.kind == "maxpool":
                dst = bufs[op.dst]
                _, H, W, Cs = src.shape
                native.check(lib.dtf_maxpool3x3(src.data_ptr(), dst.data_ptr() + 2 * op.dst_off, B, H, W, op.channels, Cs,
                                                dst.shape[3], op.stride, st), "maxpool")
            elif op.kind == "avgpool":
                dst = bufs[op.dst]
                _, H, W, Cs = src.shape
                native.check(lib.dtf_avgpool3x3_same(src.data_ptr(), dst.data_ptr(), B, H, W, op.channels, Cs, dst.shape[3], st),
                             "avgpool")
            elif op.kind == "gap":
                _, H, W, Cs = src.shape
                native.check(lib.dtf_global_avgpool(src.data_ptr(), bufs["pool_3"].data_ptr(), B, H * W, Cs, st), "gap")
            if plan is not None:
                plan.done(i)
        if plan is not None:
            plan.end()
        return bufs["pool_3"]

    def _stream_plan(self, B: int, bufs) -> Optional["StreamPlan"]:
        """The forward program over ``DTF_INCEPTION_STREAMS`` streams (default 2; 1 = one stream; batch 128: 4.78 / 4.13 / 4.38 / 4.48 ms with 1 / 2 / 3 / 4), planned once per batch
        size with the core-selection timings as cost estimates."""
        K = max(1, int(os.environ.get("DTF_INCEPTION_STREAMS", "2")))
        if K == 1 or B not in self._core:
            return None   # (core selection itself runs layer by layer on one stream)
        if B not in self._plans:
            costs = []
            for op in self.graph.ops:
                if op.kind == "conv":
                    s = op.conv
                    _, ho, wo, _ = bufs[op.dst].shape
                    t = self._core_us.get(B, {}).get(s.name)
                    costs.append(t if t else 4.0 + 2.0 * B * ho * wo * s.cout * s.kh * s.kw * s.cin / 350e12 * 1e6)
                else:
                    src = bufs.get(op.src)
                    costs.append(4.0 + (4.0 * src.numel() / 3e12 * 1e6 if src is not None else 0.0))
            self._plans[B] = StreamPlan(self.graph.ops, costs, K, self.device)
        return self._plans[B]

    def _select_cores(self, x: torch.Tensor, bufs: Dict[str, torch.Tensor]) -> None:
        """Per-layer choice between the two conv GEMM cores for this batch size (measured once, best of 3)."""
        from ..ops import hip_ops
        B = x.shape[0]
        if self.conv_core == "v1":
            self._core[B] = {}
            return
        hip_ops.igemm2_init()
        if self.conv_core == "v2":
            self._core[B] = {name: 2 for name in self.graph.convs}
            return
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        choice: Dict[str, int] = {}
        for op in self.graph.ops:
            if op.kind != "conv":
                continue
            s = op.conv
            src = x if op.src == "input" else bufs[op.src]
            wp, b = self.packed(s.name)
            t = {}
            for core in ((1, 2, 3, 4, 5) if s.cout > 64 else (1, 2, 3)):
                best = float("inf")
                for rep in range(4):
                    ev0.record()
                    hip_ops.conv_bias_relu(src, wp, b, Cin=s.cin, KH=s.kh, KW=s.kw, stride=s.stride, pad=s.pad, relu=True,
                                           out=bufs[op.dst], out_channel_offset=op.dst_off, in_channel_offset=op.src_off, core=core)
                    ev1.record()
                    ev1.synchronize()
                    if rep:
                        best = min(best, ev0.elapsed_time(ev1))
                t[core] = best
            fastest = min(t, key=t.get)
            choice[s.name] = fastest if t[fastest] < 0.97 * t[1] else 1
            self._core_us.setdefault(B, {})[s.name] = 1e3 * t[choice[s.name]]
        self._core[B] = choice

    def forward(self, x: torch.Tensor, use_graph: bool = True) -> torch.Tensor:
        """bf16 NHWC ``[B,299,299,8]`` -> fp32 bottlenecks ``[B,2048]`` (a fresh tensor)."""
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1:] == (299, 299, INPUT_PAD_CHANNELS) and x.is_contiguous()
        B = x.shape[0]
        bufs = self._alloc(B)
        if B not in self._core:
            self._select_cores(x, bufs)
        if not use_graph:
            return self._run_ops(x, bufs).clone()
        if B not in self._graphs:
            for s in self.graph.convs:
                self.packed(s)  # upload weights outside the capture
            self._static_in[B] = torch.zeros_like(x)
            self._static_in[B].copy_(x)
            self._run_ops(self._static_in[B], bufs)  # warm-up (lazy module loads) before capturing
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._static_out[B] = self._run_ops(self._static_in[B], bufs)
            self._graphs[B] = g
        self._static_in[B].copy_(x)
        self._graphs[B].replay()
        return self._static_out[B].clone()


# ----------------------------------------------------------------------------------- oracle
@torch.no_grad()
def reference_forward(model: InceptionV3, x_nchw: torch.Tensor, bf16_sim: bool = False,
                      return_buffers: bool = False, weights: Optional[Dict[str, tuple]] = None):
    """Plain PyTorch fp32 forward of the same op program.  ``x_nchw`` fp32 ``[B,3,299,299]`` already
    normalised.  ``bf16_sim`` rounds weights and every stored activation to bf16 like the native path.
    ``weights``: a dict the device copies of the folded filters are kept in between calls."""
    r = (lambda t: t.to(torch.bfloat16).float()) if bf16_sim else (lambda t: t)
    g = model.graph
    dev = x_nchw.device
    bufs: Dict[str, torch.Tensor] = {"input": r(x_nchw)}
    B = x_nchw.shape[0]
    for op in g.ops:
        src = bufs[op.src]
        if op.kind == "conv":
            s = op.conv
            if weights is not None and s.name in weights:
                w, b = weights[s.name]
            else:
                w, b = model.folded(s.name)
                w, b = r(w.to(dev)).permute(3, 2, 0, 1).contiguous(), b.to(dev)
                if weights is not None:
                    weights[s.name] = (w, b)
            cin = w.shape[1]
            y = F.relu(F.conv2d(src[:, op.src_off:op.src_off + cin], w, b, stride=s.stride, padding=s.pad))
            y = r(y)
        elif op.kind == "maxpool":
            y = F.max_pool2d(src, 3, op.stride)
        elif op.kind == "avgpool":
            y = r(F.avg_pool2d(src, 3, 1, 1, count_include_pad=False))
        elif op.kind == "gap":
            bufs["pool_3"] = src.mean((2, 3))
            continue
        if op.dst not in bufs:
            h, w_, c = g.shapes[op.dst]
            bufs[op.dst] = torch.zeros(B, c, h, w_, device=dev)
        bufs[op.dst][:, op.dst_off:op.dst_off + y.shape[1]] = y
    return (bufs["pool_3"], bufs) if return_buffers else bufs["pool_3"]


class Fp32Backbone:
    """Reference-precision feature extractor (``--backbone_precision fp32``): the op program of ``InceptionV3`` with fp32
    filters, fp32 stored activations and fp32 accumulation, run through the library convolutions of PyTorch-ROCm (MIOpen),
    no reduced-precision modes.  The reference computes ``pool_3/_reshape:0`` in float32 (the frozen graph it loads is a
    float32 graph, ``/root/reference/retrain1/retrain.py:73-82``, one image per ``sess.run``); the default path here stores
    activations in bf16 (``tests/test_inception_gpu.py`` bounds what that costs: about 1e-2 of the feature scale) and this one
    is for the user who wants bottleneck files of the reference's precision: 28.2 ms per batch of 128 (4.5 k images/s against
    27 k), after ~9 s of algorithm search by the library on the first batch.
    Takes the same NHWC-8 input batch as ``InceptionV3.forward`` (bf16, or fp32 from ``ImagePreprocessor(fp32=True)`` /
    ``DecodePipeline(fp32=True)``, which keep the bilinear-resized pixels unrounded)."""

    def __init__(self, model: InceptionV3):
        self.model = model
        self._weights: Dict[str, tuple] = {}

    def invalidate(self) -> None:
        """After the model's weights changed (``load_tf_weights`` / ``load_folded`` / calibration)."""
        self._weights = {}

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[1:] == (299, 299, INPUT_PAD_CHANNELS), tuple(x.shape)
        x = x[..., :3].permute(0, 3, 1, 2).float().contiguous()
        conv_tf32, mm_tf32 = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
        torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
        try:
            return reference_forward(self.model, x, weights=self._weights)
        finally:
            torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = conv_tf32, mm_tf32


@torch.no_grad()
def calibrate_batchnorm_reference(model: InceptionV3, x: torch.Tensor) -> None:
    """``inception_train.calibrate_batchnorm`` in plain PyTorch fp32 (runs on the CPU too): every conv's moving statistics
    become the batch statistics of the calibration batch ``x`` (NHWC-8 like ``forward``'s input), layer by layer, so that a
    RANDOM-INIT backbone keeps sane activation scales.  Pretrained weights need none of this."""
    g = model.graph
    x = x[..., :3].permute(0, 3, 1, 2).float().contiguous()
    dev = x.device
    bufs: Dict[str, torch.Tensor] = {"input": x}
    B = x.shape[0]
    for op in g.ops:
        src = bufs[op.src]
        if op.kind == "conv":
            s = op.conv
            p = model.params[s.name]
            w = p["weights"].to(dev).permute(3, 2, 0, 1).contiguous()
            raw = F.conv2d(src[:, op.src_off:op.src_off + w.shape[1]], w, None, stride=s.stride, padding=s.pad)
            var, mean = torch.var_mean(raw, dim=(0, 2, 3), unbiased=False)
            p["moving_mean"], p["moving_variance"] = mean.cpu(), var.cpu()
            scale = (p["gamma"].to(dev) / torch.sqrt(var + BN_EPS)).view(1, -1, 1, 1)
            y = F.relu((raw - mean.view(1, -1, 1, 1)) * scale + p["beta"].to(dev).view(1, -1, 1, 1))
        elif op.kind == "maxpool":
            y = F.max_pool2d(src, 3, op.stride)
        elif op.kind == "avgpool":
            y = F.avg_pool2d(src, 3, 1, 1, count_include_pad=False)
        else:
            continue
        if op.dst not in bufs:
            h, w_, c = g.shapes[op.dst]
            bufs[op.dst] = torch.zeros(B, c, h, w_, device=dev)
        bufs[op.dst][:, op.dst_off:op.dst_off + y.shape[1]] = y
    model._packed.clear()
