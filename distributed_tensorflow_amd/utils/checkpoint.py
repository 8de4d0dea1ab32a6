"""Checkpoint save / restore keyed by the reference's TensorFlow variable names (components C25, C26).

Reference behaviour: ``tf.train.Saver()`` over all global variables — weights, Adam slots
(``<var>/Adam``, ``<var>/Adam_1``), ``beta1_power``/``beta2_power`` and (demo2) the ``global_step``
— saved once at the end as ``model/train.ckpt`` (``demo1/train.py:144,165``) or periodically by the
Supervisor as ``logs/model.ckpt-<global_step>`` (``demo2/train.py:166-172``) with a ``checkpoint``
state file naming the newest one; ``demo2/test.py:182`` hard-codes ``model.ckpt-3706`` (Q21) where
this module resolves the latest checkpoint instead.

Storage is the native CRC-protected container (``csrc/host/checkpoint.cc``; atomic tmp+rename so a
crash during the timed autosave cannot leave a torn file).  Because the engines keep parameters in
the TF layouts (HWIO filters, [in,out] dense) the export needs no permutation.
``export_npz`` writes the same tensors as a plain ``.npz`` for consumers without this package.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import re
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np

from .. import native
from ..models import mnist_cnn

_DTYPES = [np.dtype("float32"), np.dtype("float64"), np.dtype("int32"), np.dtype("int64"),
           np.dtype("uint8"), np.dtype("uint16"), np.dtype("float16"), np.dtype("int8"), np.dtype("bool")]
_DT_CODE = {d: i for i, d in enumerate(_DTYPES)}
BF16_CODE = 100  # stored as uint16 bit patterns


class CheckpointError(RuntimeError):
    pass


def _to_numpy(v):
    try:
        import torch
        if isinstance(v, torch.Tensor):
            t = v.detach().cpu()
            if t.dtype == torch.bfloat16:
                return t.view(torch.int16).numpy().view(np.uint16), BF16_CODE
            return t.numpy(), None
    except ImportError:
        pass
    return np.asarray(v), None


def save_tensors(path: str, tensors: Dict[str, object]) -> str:
    """Write ``name -> tensor`` atomically.  Returns ``path``."""
    lib = native.host()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    h = lib.dtf_ckpt_begin(path.encode())
    if not h:
        raise CheckpointError("cannot open %s for writing" % path)
    rc = 0
    for name, v in tensors.items():
        a, code = _to_numpy(v)
        a = np.ascontiguousarray(a)
        if code is None:
            if a.dtype not in _DT_CODE:
                raise CheckpointError("unsupported dtype %s for %s" % (a.dtype, name))
            code = _DT_CODE[a.dtype]
        dims = (C.c_uint64 * max(1, a.ndim))(*a.shape)
        rc = rc or lib.dtf_ckpt_add(h, name.encode(), code, a.ndim, dims, a.ctypes.data_as(C.c_void_p), a.nbytes)
    rc2 = lib.dtf_ckpt_finish(h)
    if rc or rc2:
        raise CheckpointError("writing %s failed (%d/%d)" % (path, rc, rc2))
    return path


def load_tensors(path: str) -> "OrderedDict[str, np.ndarray]":
    """Read every tensor (CRC verified).  bf16 entries come back as ``uint16`` bit patterns."""
    lib = native.host()
    h = lib.dtf_ckpt_open(path.encode())
    if not h:
        raise CheckpointError("%s is not a readable checkpoint" % path)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    try:
        name = C.create_string_buffer(1024)
        dt, nd, nbytes = C.c_int(), C.c_int(), C.c_uint64()
        dims = (C.c_uint64 * 8)()
        for i in range(lib.dtf_ckpt_count(h)):
            native.check(lib.dtf_ckpt_entry(h, i, name, 1024, C.byref(dt), C.byref(nd), dims, C.byref(nbytes)), "ckpt entry")
            dtype = np.dtype("uint16") if dt.value == BF16_CODE else _DTYPES[dt.value]
            shape = tuple(int(dims[d]) for d in range(nd.value))
            a = np.empty(shape, dtype)
            if a.nbytes != nbytes.value:
                raise CheckpointError("size mismatch for %s" % name.value.decode())
            rc = lib.dtf_ckpt_read(h, i, a.ctypes.data_as(C.c_void_p))
            if rc == -74:
                raise CheckpointError("CRC mismatch in %s (%s)" % (path, name.value.decode()))
            native.check(rc, "ckpt read")
            out[name.value.decode()] = a
    finally:
        lib.dtf_ckpt_close(h)
    return out


# ----------------------------------------------------------------------------- TensorFlow tensor bundle (tf.train.Saver's files)
_TF_DTYPE = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.uint8): 4,
             np.dtype(np.int64): 9, np.dtype(np.bool_): 10}
_NP_OF_TF = {v: k for k, v in _TF_DTYPE.items()}


def save_tf_bundle(prefix: str, tensors: Dict[str, object]) -> str:
    """``<prefix>.index`` + ``<prefix>.data-00000-of-00001``: the checkpoint files ``tf.train.Saver().save(sess, prefix)``
    writes (``demo1/train.py:144,165``), produced by the native writer (``csrc/host/tensor_bundle.cc``) — an SSTable of
    ``BundleEntryProto`` records over the raw tensor bytes, CRC32C per block and per tensor.  Keys are the variables' TF
    names.  Atomic (tmp + rename)."""
    lib = native.host()
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or ".", exist_ok=True)
    h = lib.dtf_bundle_begin(prefix.encode())
    if not h:
        raise CheckpointError("cannot start tensor bundle " + prefix)
    for name, v in tensors.items():
        a = np.asarray(_to_numpy(v)[0], order="C")  # (ascontiguousarray would promote a scalar to 1-d)
        if a.dtype not in _TF_DTYPE:
            raise CheckpointError("%s: dtype %s has no TensorFlow DataType here" % (name, a.dtype))
        dims = (C.c_int64 * max(1, a.ndim))(*a.shape)
        if lib.dtf_bundle_add(h, name.encode(), _TF_DTYPE[a.dtype], a.ndim, dims, a.ctypes.data_as(C.c_void_p), a.nbytes) != 0:
            raise CheckpointError("tensor bundle: cannot add " + name)
    if lib.dtf_bundle_finish(h) != 0:
        raise CheckpointError("tensor bundle: writing %s failed" % prefix)
    return prefix


def load_tf_bundle(prefix: str) -> "OrderedDict[str, np.ndarray]":
    """Read a tensor bundle (ours or one written by TensorFlow's ``Saver``: any uncompressed single-shard table) without
    TensorFlow; per-block and per-tensor CRCs are verified."""
    lib = native.host()
    h = lib.dtf_bundle_open(prefix.encode())
    if not h:
        raise CheckpointError("%s.index: not a readable single-shard tensor bundle" % prefix)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    try:
        name = C.create_string_buffer(1024)
        dtype, ndim, nbytes = C.c_int(), C.c_int(), C.c_uint64()
        dims = (C.c_int64 * 8)()
        for i in range(lib.dtf_bundle_count(h)):
            if lib.dtf_bundle_entry(h, i, name, 1024, C.byref(dtype), C.byref(ndim), dims, C.byref(nbytes)) != 0:
                raise CheckpointError("tensor bundle: bad entry %d" % i)
            if dtype.value not in _NP_OF_TF:
                continue  # (strings / resources: nothing this framework stores)
            a = np.empty(tuple(dims[d] for d in range(ndim.value)), _NP_OF_TF[dtype.value])
            if a.nbytes != nbytes.value:
                raise CheckpointError("%s: %d bytes for shape %s" % (name.value.decode(), nbytes.value, a.shape))
            rc = lib.dtf_bundle_read(h, i, a.ctypes.data_as(C.c_void_p))
            if rc != 0:
                raise CheckpointError("%s: %s" % (name.value.decode(), "CRC mismatch" if rc == -74 else "read error %d" % rc))
            out[name.value.decode()] = a
    finally:
        lib.dtf_bundle_close(h)
    return out


def is_tf_bundle(prefix: str) -> bool:
    return os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")


# ----------------------------------------------------------------------------- TF-name layout
def mnist_tf_names(with_global_step: bool) -> "OrderedDict[str, str]":
    """engine parameter name -> TF variable name."""
    return OrderedDict((k, spec[0]) for k, spec in mnist_cnn.PARAM_SPECS.items())


def mnist_state_to_tf(engine, *, with_global_step: bool = True, beta1: float = 0.9, beta2: float = 0.999
                      ) -> "OrderedDict[str, np.ndarray]":
    """Everything ``tf.train.Saver()`` would have saved for the MNIST graph, under the TF names."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    sd = engine.state_dict()
    step = int(np.asarray(sd["step"].cpu()).reshape(-1)[0])
    P, M, V = (engine.param_views(sd[k]) for k in ("P", "M", "V"))
    for k, (tf_name, _shape, _kind) in mnist_cnn.PARAM_SPECS.items():
        out[tf_name] = P[k].detach().cpu().numpy()
        out[tf_name + "/Adam"] = M[k].detach().cpu().numpy()
        out[tf_name + "/Adam_1"] = V[k].detach().cpu().numpy()
    # TF keeps beta^(t+1) after t updates (initial value beta, multiplied after every apply)
    out["train_step/beta1_power"] = np.asarray(beta1 ** (step + 1), np.float32)
    out["train_step/beta2_power"] = np.asarray(beta2 ** (step + 1), np.float32)
    if with_global_step:
        out["Variable"] = np.asarray(step, np.int32)
    else:
        out["_dtf/step"] = np.asarray(step, np.int32)
    return out


def mnist_state_from_tf(engine, tensors: Dict[str, np.ndarray]) -> int:
    """Load a TF-named bundle back into an engine; returns the restored step."""
    import torch
    sd = engine.state_dict()
    P, M, V = (engine.param_views(sd[k]) for k in ("P", "M", "V"))
    for k, (tf_name, shape, _kind) in mnist_cnn.PARAM_SPECS.items():
        if tf_name not in tensors:
            raise CheckpointError("variable %s missing from checkpoint" % tf_name)
        P[k].copy_(torch.from_numpy(np.asarray(tensors[tf_name], np.float32).reshape(shape)))
        if tf_name + "/Adam" in tensors:
            M[k].copy_(torch.from_numpy(np.asarray(tensors[tf_name + "/Adam"], np.float32).reshape(shape)))
            V[k].copy_(torch.from_numpy(np.asarray(tensors[tf_name + "/Adam_1"], np.float32).reshape(shape)))
    step = int(np.asarray(tensors.get("Variable", tensors.get("_dtf/step", 0))).reshape(-1)[0])
    sd["step"] = torch.tensor([step], dtype=torch.int32)
    engine.load_state_dict(sd)
    return step


# ----------------------------------------------------------------------------- Saver-like front end
class Saver:
    """``saver.save(sess, path[, global_step])`` / ``saver.restore(sess, path)`` for an engine."""

    def __init__(self, engine, *, with_global_step: bool = True, max_to_keep: int = 5, tf_bundle: bool = True):
        self.engine = engine
        self.with_global_step = with_global_step
        self.max_to_keep = max_to_keep
        # also write what tf.train.Saver writes, <path>.index + <path>.data-00000-of-00001 (TensorFlow can restore it;
        # restore() reads either form)
        self.tf_bundle = tf_bundle
        self._kept = []

    def save(self, save_path: str, global_step: Optional[int] = None) -> str:
        tensors = mnist_state_to_tf(self.engine, with_global_step=self.with_global_step)
        if global_step is not None:
            # name the file after the step INSIDE the snapshot: the host's mirror of the step counter can be up to one
            # multi-step graph launch ahead of what the stream-ordered snapshot saw
            global_step = int(np.asarray(tensors.get("Variable", tensors.get("_dtf/step", global_step))).reshape(-1)[0])
        path = save_path if global_step is None else "%s-%d" % (save_path, global_step)
        save_tensors(path, tensors)
        if self.tf_bundle:
            # TF has no "_dtf/step": the demo1 graph keeps no global step, so that entry stays in the native container only
            save_tf_bundle(path, OrderedDict((k, v) for k, v in tensors.items() if not k.startswith("_dtf/")))
        update_checkpoint_state(os.path.dirname(os.path.abspath(path)) or ".", path)
        self._kept.append(path)
        while self.max_to_keep and len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            for f in (old, old + ".index", old + ".data-00000-of-00001"):
                if old != path and os.path.exists(f):
                    os.remove(f)
        return path

    def restore(self, path: str) -> int:
        """``path``: the native container, or the prefix of a TensorFlow tensor bundle (``model/train.ckpt`` as written by
        ``tf.train.Saver``: ``demo1/test.py:182`` restores exactly that)."""
        if not os.path.isfile(path) and is_tf_bundle(path):
            return mnist_state_from_tf(self.engine, load_tf_bundle(path))
        return mnist_state_from_tf(self.engine, load_tensors(path))


def update_checkpoint_state(directory: str, latest_path: str) -> None:
    """The ``checkpoint`` state file TF writes next to checkpoints."""
    tmp = os.path.join(directory, "checkpoint.tmp")
    with open(tmp, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(latest_path))
    os.replace(tmp, os.path.join(directory, "checkpoint"))


def latest_checkpoint(directory: str, prefix: str = "model.ckpt") -> Optional[str]:
    """Newest checkpoint in ``directory`` (state file first, else highest ``-<step>`` suffix)."""
    state = os.path.join(directory, "checkpoint")
    if os.path.exists(state):
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
        if m:
            p = os.path.join(directory, m.group(1))
            if os.path.exists(p):
                return p
    best, best_step = None, -1
    for p in glob.glob(os.path.join(directory, prefix + "-*")):
        m = re.search(r"-(\d+)$", p)
        if m and int(m.group(1)) > best_step:
            best, best_step = p, int(m.group(1))
    return best


def export_npz(path: str, tensors: Dict[str, np.ndarray]) -> str:
    np.savez(path, **{k.replace("/", "__"): v for k, v in tensors.items()})
    return path
