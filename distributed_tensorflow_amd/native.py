"""ctypes bindings to the in-tree native libraries.

* ``hip()``  -> ``_lib/libdtf_hip.so``  (gfx950 kernels + step executor; needs a GPU to *run*, loads anywhere ROCm is installed)
* ``host()`` -> ``_lib/libdtf_host.so`` (CRC32C, tfevents writer, rendezvous store, checkpoint container)

There is deliberately no silent fallback: if a library is missing it is built on the spot
(hipcc cross-compiles without a GPU) and if that fails the import error is raised — on a GPU
box the HIP path must be the one that runs.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib")
_lock = threading.Lock()
_hip: Optional[C.CDLL] = None
_host: Optional[C.CDLL] = None

c_void_p, c_int, c_long, c_float, c_uint64, c_int64, c_char_p = (
    C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_int64, C.c_char_p)


class NativeError(RuntimeError):
    pass


# ----------------------------------------------------------------------------- struct mirrors (csrc/hip/dtf_structs.h)
class HeadParams(C.Structure):
    _fields_ = [
        ("slabs", c_void_p), ("nslab", c_int), ("slab_stride", c_long),
        ("b_fc1", c_void_p), ("W_fc2", c_void_p), ("b_fc2", c_void_p),
        ("labels_all", c_void_p), ("perm", c_void_p), ("step_ptr", c_void_p),
        ("spe", c_int), ("B", c_int), ("keep_prob", c_float), ("grad_scale", c_float),
        ("seed", C.c_ulonglong), ("double_softmax", c_int), ("training", c_int),
        ("h_drop", c_void_p), ("dh", c_void_p), ("dz", c_void_p),
        ("db_fc2", c_void_p), ("db_fc1", c_void_p), ("metrics", c_void_p), ("metrics_ring", c_int),
        ("probs", c_void_p), ("pred", c_void_p), ("zero_buf", c_void_p), ("zero_n4", c_long),
        ("f32", c_int),
    ]


class OptimParams(C.Structure):
    _fields_ = [
        ("P", c_void_p), ("G", c_void_p), ("M", c_void_p), ("V", c_void_p),
        ("n", c_long), ("zero_from", c_long),
        ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("grad_scale", c_float),
        ("step", c_void_p), ("ticket", c_void_p), ("metrics_ring", c_void_p), ("ring", c_int),
        ("Gsrc", c_void_p), ("Gsrcs", c_void_p * 8), ("nsrc", c_int), ("skip", c_void_p),
        ("Wsh", c_void_p * 16), ("wsh_world", c_int), ("wsh_rank", c_int),  # ("wshard" schedule: set by engine.hip only)
    ]


class MnistEngineStruct(C.Structure):  # csrc/hip/engine.hip: struct MnistEngine
    _fields_ = [
        ("B", c_int), ("maxB", c_int), ("spe", c_int), ("world", c_int), ("double_softmax", c_int),
        ("fc1_splits", c_int), ("bwd_splits", c_int), ("dw2_splits", c_int), ("head_blocks", c_int), ("ring", c_int),
        ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float), ("keep_prob", c_float),
        ("seed", C.c_ulonglong),
        ("images", c_void_p), ("labels", c_void_p), ("perm", c_void_p),
        ("P", c_void_p), ("G", c_void_p), ("M", c_void_p), ("V", c_void_p),
        ("n_arena", c_long),
        ("off_Wfc1", c_long), ("off_bfc1", c_long), ("off_Wfc2", c_long), ("off_bfc2", c_long),
        ("off_Wc2", c_long), ("off_bc2", c_long), ("off_Wc1", c_long), ("off_bc1", c_long),
        ("step", c_void_p), ("ticket", c_void_p), ("metrics_ring", c_void_p),
        ("pool1", c_void_p), ("amax1", c_void_p), ("pool2", c_void_p), ("amax2", c_void_p),
        ("slabs", c_void_p), ("h_drop", c_void_p), ("dh", c_void_p), ("dz", c_void_p),
        ("dpool2", c_void_p), ("dy2", c_void_p), ("dpool1", c_void_p),
        ("G_fc1_dst", c_void_p), ("G_src", c_void_p),
        ("gather_A", c_void_p * 8), ("gather_B", c_void_p * 8), ("gather_comm", c_void_p), ("gather_n", c_int), ("gather_rank", c_int),
        ("wc2f", c_void_p), ("wc2b", c_void_p),
        ("gather_local", c_void_p), ("gather_bytes", c_long),
        ("f32", c_int), ("skip", c_void_p),
        ("shard_comm", c_void_p), ("shard_n", c_int), ("shard_lo", c_int), ("shard_w", c_int),
        ("wshard", c_int), ("wshard_rows_a", c_int), ("early_push", c_int),
    ]


def _ensure_built(name: str) -> str:
    path = os.path.join(_LIBDIR, name)
    from . import build
    which = "hip" if "hip" in name else "host"
    stale = os.path.exists(path) and os.path.isdir(build.CSRC) and not build.is_current(which)
    if stale and os.environ.get("DTF_ALLOW_STALE_LIB", "0") == "1":
        stale = False
    if not os.path.exists(path) or stale:
        # (a library that was not built from the sources next to it is rebuilt, never used silently)
        (build.build_hip if which == "hip" else build.build_host)()
    if not os.path.exists(path):
        raise NativeError("native library %s is missing and could not be built" % path)
    return path


def _sig(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


def hip() -> C.CDLL:
    global _hip
    with _lock:
        if _hip is not None:
            return _hip
        # torch first: it ships its own libamdhip64, and the process must only ever have ONE HIP runtime.  Loaded before
        # torch, this library binds the system runtime, torch then brings a second one, and the first kernel-attribute
        # call fails with hipErrorNoDevice (seen when build() and smoke() ran in one process).
        import torch  # noqa: F401
        lib = C.CDLL(_ensure_built("libdtf_hip.so"))
        P, I, L, F = c_void_p, c_int, c_long, c_float
        _sig(lib, "dtf_version", I, [])
        _sig(lib, "dtf_hip_source_digest", c_char_p, [])
        _sig(lib, "dtf_fc_fwd", I, [P, I, P, P, I, I, I, I, P, I, P])
        _sig(lib, "dtf_fc_bwd_data", I, [P, P, P, I, I, I, I, P])
        _sig(lib, "dtf_fc_dw", I, [P, I, P, P, I, I, I, I, P])
        _sig(lib, "dtf_fc_dw_pair", I, [P, P, P, I, I, P, P, P, I, I, I, P])
        _sig(lib, "dtf_conv2_lds_init", I, [])
        _sig(lib, "dtf_conv2_fwd_lds", I, [P, P, P, P, P, I, P])
        _sig(lib, "dtf_tensor_stats", I, [P, C.c_long, P, I, P, P, P])
        L = C.c_long
        _sig(lib, "dtf_conv_fwd_stats", I, [P, P, P, P] + [I] * 13 + [P])
        _sig(lib, "dtf_conv_bwd_data", I, [P, P, P] + [I] * 14 + [P])
        _sig(lib, "dtf_conv_dw_general", I, [P, I, P, P] + [I] * 13 + [P])
        _sig(lib, "dtf_bn_finalize", I, [P, P, P, P, P, P, I, L, F, F, P])
        _sig(lib, "dtf_bn_relu_fwd", I, [P, P, P, L, I, I, P])
        _sig(lib, "dtf_bn_finalize_relu_fwd", I, [P, P, P, P, P, P, P, P, P, L, I, I, F, F, P])
        _sig(lib, "dtf_bn_relu_bwd_reduce", I, [P, I, P, P, P, L, I, P])
        _sig(lib, "dtf_bn_relu_bwd_apply", I, [P, I, P, P, P, P, P, P, L, I, P])
        _sig(lib, "dtf_maxpool3x3_arg", I, [P, P, P] + [I] * 7 + [P])
        _sig(lib, "dtf_maxpool3x3_bwd", I, [P, I, P, P, I] + [I] * 6 + [P])
        _sig(lib, "dtf_avgpool3x3_same_bwd", I, [P, I, P, I] + [I] * 5 + [P])
        _sig(lib, "dtf_global_avgpool_bwd", I, [P, P, I, I, I, P])
        _sig(lib, "dtf_igemm2_init", I, [])
        _sig(lib, "dtf_conv_fwd_stats_v2", I, [P, P, P, P] + [I] * 14 + [P])
        _sig(lib, "dtf_conv_bias_relu_v2", I, [P, P, P, P] + [I] * 16 + [P])
        _sig(lib, "dtf_conv_bwd_data_v2", I, [P, P, P] + [I] * 15 + [P])
        _sig(lib, "dtf_pack_conv_weights", I, [P, P, I, P, P, P])
        _sig(lib, "dtf_igemm2_dw_init", I, [])
        _sig(lib, "dtf_conv_dw_v2", I, [P, I, P, P] + [I] * 14 + [P])
        _sig(lib, "dtf_conv2_bwd_lds_init", I, [])
        _sig(lib, "dtf_conv2_bwd_lds", I, [P, P, P, P, P, P, P, P, I, I, P])
        _sig(lib, "dtf_conv_bias_relu_pool", I, [P, P, P, P, P, I, I, I, I, I, I, I, P])
        _sig(lib, "dtf_conv_bwd_data_mask", I, [P, P, P, P, I, I, I, I, I, I, I, P])
        _sig(lib, "dtf_conv_dw", I, [P, P, P, I, I, I, I, I, I, I, I, P])
        _sig(lib, "dtf_conv_bias_relu", I, [P, P, P, P] + [I] * 15 + [P])
        _sig(lib, "dtf_conv1_fwd", I, [P, P, P, I, I, P, P, P, P, P])
        _sig(lib, "dtf_conv1_dw", I, [P, P, P, I, I, P, P, P, P, P])
        _sig(lib, "dtf_unpool", I, [P, P, P, P, P, I, I, I, I, P])
        _sig(lib, "dtf_head", I, [C.POINTER(HeadParams), I, P])
        _sig(lib, "dtf_adam_tf", I, [C.POINTER(OptimParams), P])
        _sig(lib, "dtf_sgd", I, [C.POINTER(OptimParams), P])
        _sig(lib, "dtf_cast_f32_bf16", I, [P, P, C.c_long, P])
        _sig(lib, "dtf_cast_bf16_f32", I, [P, P, C.c_long, P])
        _sig(lib, "dtf_mnist_fwd_bwd", I, [C.POINTER(MnistEngineStruct), I, P])
        _sig(lib, "dtf_mnist_apply", I, [C.POINTER(MnistEngineStruct), P])
        _sig(lib, "dtf_mnist_apply_part", I, [C.POINTER(MnistEngineStruct), I, P])
        _sig(lib, "dtf_mnist_apply_part_ipc", I, [C.POINTER(MnistEngineStruct), I, P, P])
        _sig(lib, "dtf_mnist_wshard_load", I, [C.POINTER(MnistEngineStruct), P])
        _sig(lib, "dtf_mnist_wshard_store", I, [C.POINTER(MnistEngineStruct), P])
        _sig(lib, "dtf_mnist_eval", I, [C.POINTER(MnistEngineStruct), P, P, I, P, P, P, P])
        # exact-fp32 (reference precision) kernels: f32_fc.hip, f32_conv2.hip
        _sig(lib, "dtf_fc_fwd_f32", I, [P, P, P, I, I, I, I, P, I, P])
        _sig(lib, "dtf_fc_bwd_data_f32", I, [P, P, P, I, I, I, I, P])
        _sig(lib, "dtf_fc_dw_f32", I, [P, I, P, I, P, I, I, I, I, P])
        _sig(lib, "dtf_conv2_f32_init", I, [])
        _sig(lib, "dtf_conv1_fwd_f32", I, [P, P, P, I, I, P, P, P, P, P])
        _sig(lib, "dtf_conv2_fwd_f32", I, [P, P, P, P, P, I, P])
        _sig(lib, "dtf_conv2_bwd_data_f32", I, [P, P, P, P, P, P, I, C.POINTER(OptimParams), P])
        _sig(lib, "dtf_conv_dw_duo_f32", I, [P, P, P, P, P, P, I, P, P, P, I, P, P, P, P, C.POINTER(OptimParams), P])
        # fp32 on the bf16 matrix cores ("bf16x6": csrc/hip/bf16x3.h, f32x3_conv2.hip).  DTF_F32_EMULATION=0 -> exact-fp32 MFMA
        _sig(lib, "dtf_f32_emulation_set", None, [I])
        _sig(lib, "dtf_f32_emulation", I, [])
        _sig(lib, "dtf_conv2_pack3", I, [P, P, P, P])
        _sig(lib, "dtf_conv2_fwd_x3_packed", I, [P, P, P, P, P, I, P])
        _sig(lib, "dtf_conv2_bwd_data_x3_packed", I, [P, P, P, P, P, P, I, P])
        _sig(lib, "dtf_conv2_fwd_x3", I, [P, P, P, P, P, I, P])
        _sig(lib, "dtf_conv2_bwd_data_x3", I, [P, P, P, P, P, P, I, P])
        _sig(lib, "dtf_conv1_fwd_f32_pack3", I, [P, P, P, I, I, P, P, P, P, P, P, P, P])
        lib.dtf_f32_emulation_set(0 if os.environ.get("DTF_F32_EMULATION", "1").lower() in ("0", "off", "false", "no") else 1)
        _sig(lib, "dtf_graph_begin", I, [P])
        _sig(lib, "dtf_graph_end", I, [P, C.POINTER(c_void_p)])
        _sig(lib, "dtf_graph_launch", I, [P, P])
        _sig(lib, "dtf_graph_num_nodes", I, [P])
        _sig(lib, "dtf_graph_free", None, [P])
        # Inception / retrain ops (csrc/hip/inception_ops.hip, retrain_head.hip).  Every exported function gets
        # an explicit signature here: a missing one would make ctypes truncate 64-bit pointers to c_int.
        _sig(lib, "dtf_preprocess_images", I, [P, P, P, I, I, P])
        _sig(lib, "dtf_preprocess_images_f32", I, [P, P, P, I, I, P])
        _sig(lib, "dtf_maxpool3x3", I, [P, P] + [I] * 7 + [P])
        _sig(lib, "dtf_avgpool3x3_same", I, [P, P] + [I] * 6 + [P])
        _sig(lib, "dtf_global_avgpool", I, [P, P, I, I, I, P])
        _sig(lib, "dtf_sample_rows", I, [P, P, I, I, C.c_ulonglong, C.c_ulonglong, P, P])
        _sig(lib, "dtf_sample_rows_dev", I, [P, P, I, I, C.c_ulonglong, P, P, P])
        _sig(lib, "dtf_gather_rows", I, [P, P, P, C.c_long, I, P, P, P])
        _sig(lib, "dtf_sizeof_retrain_head_params", I, [])
        _sig(lib, "dtf_retrain_head_fwd", I, [P, P])
        _sig(lib, "dtf_retrain_head_bwd", I, [P, P, P, I, I, P, P, P, P, F, P])
        _sig(lib, "dtf_retrain_head_dx", I, [P, P, P, I, I, P, P])
        for extra in _HIP_EXTRA_SIGS:
            if hasattr(lib, extra[0]):
                _sig(lib, *extra)
        # the ctypes mirrors must be exactly as large as the structs the kernels take by value
        sizes = (C.c_int * 3)()
        _sig(lib, "dtf_struct_sizes", None, [P])(sizes)
        for mirror, native_size in zip((HeadParams, OptimParams, MnistEngineStruct), sizes):
            if C.sizeof(mirror) != native_size:
                raise NativeError("native.%s is %d bytes, the library's struct %d: the ctypes mirror is out of date" % (
                    mirror.__name__, C.sizeof(mirror), native_size))
        _hip = lib
        return lib


def f32_emulation_note() -> Optional[str]:
    """What the fp32 path runs on the bf16 matrix cores (``csrc/hip/bf16x3.h``), for bench records; None = nothing
    (``DTF_F32_EMULATION=0``: every GEMM-shaped kernel on the exact-fp32 MFMA)."""
    if not hip().dtf_f32_emulation():
        return None
    return ("bf16x6: " + ", ".join(F32_EMULATED_KERNELS) + " take each fp32 operand as three bf16 planes and each product as "
            "six v_mfma_f32_16x16x32_bf16 (fp32 accumulate); same float64-oracle tolerances as the exact-fp32 MFMA kernels "
            "(tests/test_f32_gpu.py, 1e-5 per kernel, 1e-4 step gradients); conv1, the conv filter gradients, the loss head and "
            "Adam are plain fp32")


F32_EMULATED_KERNELS = ("conv2 forward", "conv2 data gradient", "fc1 forward", "fc1 data gradient",
                        "fc1 / fc2 filter gradients")


# signatures of optional kernels (Inception pools / image ops / stats / retrain head) registered by
# the modules that own them so this file does not have to know every op
_HIP_EXTRA_SIGS = []


def register_hip_sig(name, restype, argtypes) -> None:
    _HIP_EXTRA_SIGS.append((name, restype, argtypes))
    if _hip is not None and hasattr(_hip, name):
        _sig(_hip, name, restype, argtypes)


def host() -> C.CDLL:
    global _host
    with _lock:
        if _host is not None:
            return _host
        lib = C.CDLL(_ensure_built("libdtf_host.so"))
        P, I = c_void_p, c_int
        _sig(lib, "dtf_host_source_digest", c_char_p, [])
        _sig(lib, "dtf_crc32c", C.c_uint32, [P, c_uint64])
        _sig(lib, "dtf_masked_crc32c", C.c_uint32, [P, c_uint64])
        _sig(lib, "dtf_events_open", P, [c_char_p])
        _sig(lib, "dtf_events_begin", None, [P, c_int64, C.c_double])
        _sig(lib, "dtf_events_scalar", None, [P, c_char_p, c_float])
        _sig(lib, "dtf_events_histogram", None, [P, c_char_p] + [C.c_double] * 5 + [I, P, P])
        _sig(lib, "dtf_events_end", None, [P])
        _sig(lib, "dtf_events_graph", None, [P, P, c_uint64])
        _sig(lib, "dtf_events_flush", None, [P])
        _sig(lib, "dtf_events_num_records", c_uint64, [P])
        _sig(lib, "dtf_events_close", None, [P])
        _sig(lib, "dtf_store_server_start", P, [I])
        _sig(lib, "dtf_store_server_port", I, [P])
        _sig(lib, "dtf_store_server_stop", None, [P])
        _sig(lib, "dtf_store_connect", P, [c_char_p, I, I])
        _sig(lib, "dtf_store_set", c_int64, [P, c_char_p, P, c_uint64])
        _sig(lib, "dtf_store_get", c_int64, [P, c_char_p, P, c_uint64, c_int64])
        _sig(lib, "dtf_store_add", c_int64, [P, c_char_p, c_int64])
        _sig(lib, "dtf_store_check", c_int64, [P, c_char_p])
        _sig(lib, "dtf_store_delete", c_int64, [P, c_char_p])
        _sig(lib, "dtf_store_close", None, [P])
        _sig(lib, "dtf_ckpt_begin", P, [c_char_p])
        _sig(lib, "dtf_ckpt_add", I, [P, c_char_p, I, I, P, P, c_uint64])
        _sig(lib, "dtf_ckpt_finish", I, [P])
        _sig(lib, "dtf_ckpt_open", P, [c_char_p])
        _sig(lib, "dtf_ckpt_count", I, [P])
        _sig(lib, "dtf_ckpt_entry", I, [P, I, P, I, P, P, P, P])
        _sig(lib, "dtf_ckpt_read", I, [P, I, P])
        _sig(lib, "dtf_ckpt_close", None, [P])
        _sig(lib, "dtf_bundle_begin", P, [c_char_p])
        _sig(lib, "dtf_bundle_add", I, [P, c_char_p, I, I, P, P, c_uint64])
        _sig(lib, "dtf_bundle_finish", I, [P])
        _sig(lib, "dtf_bundle_open", P, [c_char_p])
        _sig(lib, "dtf_bundle_count", I, [P])
        _sig(lib, "dtf_bundle_entry", I, [P, I, P, I, P, P, P, P])
        _sig(lib, "dtf_bundle_read", I, [P, I, P])
        _sig(lib, "dtf_bundle_close", None, [P])
        _sig(lib, "dtf_parse_floats", c_int64, [c_char_p, c_uint64, P, c_uint64])
        _sig(lib, "dtf_format_floats", c_int64, [P, c_uint64, P, c_uint64])
        _host = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeError("%s failed with code %d%s" % (
            what, rc, " (shape/alignment precondition violated)" if rc == -22 else ""))


def ptr(t) -> int:
    """Device/host address of a torch tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
