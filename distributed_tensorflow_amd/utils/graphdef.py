"""TensorFlow-free ``GraphDef`` (.pb) constant reader / writer (components C26, C28).

The reference imports ``classify_image_graph_def.pb`` with ``tf.import_graph_def`` and pulls tensors out
by name (``retrain1/retrain.py:66-74``), and freezes its retrained graph with
``convert_variables_to_constants`` (``:470-473``).  No TensorFlow exists on an MI355X box, so this
module speaks the protobuf wire format directly — only the handful of messages needed:

    GraphDef   { repeated NodeDef node = 1; }
    NodeDef    { string name = 1; string op = 2; repeated string input = 3; map<string, AttrValue> attr = 5; }
    AttrValue  { bytes s = 2; int64 i = 3; float f = 4; bool b = 5; DataType type = 6; TensorProto tensor = 8; }
    TensorProto{ DataType dtype = 1; TensorShapeProto tensor_shape = 2; bytes tensor_content = 4;
                 repeated float float_val = 5; repeated double double_val = 6; repeated int32 int_val = 7;
                 repeated int64 int64_val = 10; }
    TensorShapeProto { repeated Dim dim = 2 { int64 size = 1; } }

``read_graph_consts`` streams the node list and materialises every ``Const`` as a numpy array (the
2015 Inception graph keeps all weights as Const nodes, ~90 MB of ``tensor_content``).
``inception_weights_from_graphdef`` groups them per conv layer under the 2015 graph's names
(``<layer>/conv2d_params``, ``<layer>/batchnorm/{beta,gamma,moving_mean,moving_variance}``) in the form
``InceptionV3.load_tf_weights`` takes.  ``write_graph_consts`` is the inverse (frozen-constant export of
the retrained head next to the native bundle; also what the tests use as a fixture).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 4, 7, 9, 10
_NP_OF = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_UINT8: np.uint8, DT_INT64: np.int64,
          DT_BOOL: np.bool_}
_DT_OF = {np.dtype(v): k for k, v in _NP_OF.items()}


# ----------------------------------------------------------------------------- wire format: decode
def _varint(buf, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) triples of one message; length-delimited values are memoryviews."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("truncated length-delimited field")
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield fno, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf) -> List[int]:
    dims = []
    for fno, _wt, v in _fields(buf):
        if fno == 2:
            size = 0
            for f2, _w2, v2 in _fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
    return dims


def _packed_or_single(acc: list, wt: int, v, fmt: Optional[str], size: int) -> None:
    if wt == 2:  # packed
        if fmt is None:
            pos, n = 0, len(v)
            while pos < n:
                x, pos = _varint(v, pos)
                acc.append(_signed64(x))
        else:
            acc.extend(np.frombuffer(bytes(v), dtype=fmt).tolist())
    elif fmt is None:
        acc.append(_signed64(v))
    else:
        acc.append(struct.unpack("<" + ("f" if size == 4 else "d"), v)[0])


def parse_tensor(buf) -> np.ndarray:
    """TensorProto -> ndarray (numeric dtypes; a string tensor comes back as an object array of bytes)."""
    dtype, shape, content = DT_FLOAT, [], None
    vals: list = []
    strs: list = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dtype = v
        elif fno == 2:
            shape = _parse_shape(v)
        elif fno == 4:
            content = v
        elif fno == 5:
            _packed_or_single(vals, wt, v, "<f4", 4)
        elif fno == 6:
            _packed_or_single(vals, wt, v, "<f8", 8)
        elif fno in (7, 10, 11):
            _packed_or_single(vals, wt, v, None, 0)
        elif fno == 8:
            strs.append(bytes(v))
    if dtype == DT_STRING:
        return np.asarray(strs, dtype=object).reshape(shape) if shape else np.asarray(strs, dtype=object)
    if dtype not in _NP_OF:
        raise ValueError("unsupported tensor dtype %d" % dtype)
    npdt = np.dtype(_NP_OF[dtype])
    count = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        arr = np.frombuffer(bytes(content), dtype=npdt.newbyteorder("<")).astype(npdt, copy=False)
    else:
        arr = np.asarray(vals, dtype=npdt)
        if arr.size == 1 and count > 1:  # TF's "splat": one value repeated
            arr = np.full(count, arr[0], dtype=npdt)
        elif arr.size == 0:
            arr = np.zeros(count, dtype=npdt)
    if arr.size != count:
        raise ValueError("tensor has %d values for shape %s" % (arr.size, shape))
    return arr.reshape(shape)


def parse_node(buf) -> dict:
    node = {"name": "", "op": "", "input": [], "attr": {}}
    for fno, _wt, v in _fields(buf):
        if fno == 1:
            node["name"] = bytes(v).decode("utf-8")
        elif fno == 2:
            node["op"] = bytes(v).decode("utf-8")
        elif fno == 3:
            node["input"].append(bytes(v).decode("utf-8"))
        elif fno == 5:
            key, val = None, None
            for f2, _w2, v2 in _fields(v):
                if f2 == 1:
                    key = bytes(v2).decode("utf-8")
                elif f2 == 2:
                    val = v2
            if key is not None:
                node["attr"][key] = val  # raw AttrValue bytes, decoded lazily
    return node


def attr_value(raw) -> object:
    """Decode one AttrValue (scalar kinds + tensor)."""
    if raw is None:
        return None
    for fno, _wt, v in _fields(raw):
        if fno == 2:
            return bytes(v)
        if fno == 3:
            return _signed64(v)
        if fno == 4:
            return struct.unpack("<f", v)[0]
        if fno == 5:
            return bool(v)
        if fno == 6:
            return int(v)
        if fno == 7:
            return _parse_shape(v)
        if fno == 8:
            return parse_tensor(v)
    return None


def iter_nodes(path: str) -> Iterator[dict]:
    """Stream the NodeDefs of a serialized GraphDef (memory-mapped: a 90 MB file is not copied)."""
    data = np.memmap(path, dtype=np.uint8, mode="r")
    buf = memoryview(data)
    for fno, wt, v in _fields(buf):
        if fno == 1 and wt == 2:
            yield parse_node(v)


def read_graph_consts(path: str, dtypes=(DT_FLOAT,)) -> Dict[str, np.ndarray]:
    """name -> ndarray for every ``Const`` node whose dtype is in ``dtypes`` (None = all numeric)."""
    out: Dict[str, np.ndarray] = {}
    for node in iter_nodes(path):
        if node["op"] != "Const" or "value" not in node["attr"]:
            continue
        dt = attr_value(node["attr"].get("dtype"))
        if dtypes is not None and dt not in dtypes:
            continue
        out[node["name"]] = np.array(attr_value(node["attr"]["value"]))
    return out


def read_graph_ops(path: str) -> List[Tuple[str, str, List[str]]]:
    """(name, op, inputs) of every node — enough to check a graph's topology without TensorFlow."""
    return [(n["name"], n["op"], n["input"]) for n in iter_nodes(path)]


_BN_KEYS = ("beta", "gamma", "moving_mean", "moving_variance")


def inception_weights_from_graphdef(path: str) -> Dict[str, Dict[str, np.ndarray]]:
    """Per-layer weights of the 2015 Inception-v3 graph (``classify_image_graph_def.pb``,
    ``retrain1/retrain.py:29-33``) keyed the way ``models.inception_v3`` names its 94 convs."""
    consts = read_graph_consts(path)
    layers: Dict[str, Dict[str, np.ndarray]] = {}
    for name, arr in consts.items():
        if name.endswith("/conv2d_params"):
            layers.setdefault(name[:-len("/conv2d_params")], {})["weights"] = arr
        else:
            for k in _BN_KEYS:
                suffix = "/batchnorm/" + k
                if name.endswith(suffix):
                    layers.setdefault(name[:-len(suffix)], {})[k] = arr
    return {k: v for k, v in layers.items() if "weights" in v}


def inception_classifier_from_graphdef(path: str) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """The unused 1008-way ``softmax/{weights,biases}`` of the 2015 graph, if present."""
    consts = read_graph_consts(path)
    if "softmax/weights" in consts and "softmax/biases" in consts:
        return consts["softmax/weights"], consts["softmax/biases"]
    return None


# ----------------------------------------------------------------------------- wire format: encode
def _enc_varint(v: int) -> bytes:
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_field(fno: int, wt: int, payload) -> bytes:
    key = _enc_varint((fno << 3) | wt)
    if wt == 0:
        return key + _enc_varint(int(payload))
    if wt == 2:
        return key + _enc_varint(len(payload)) + bytes(payload)
    return key + bytes(payload)


def encode_tensor(arr: np.ndarray) -> bytes:
    arr = np.asarray(arr, order="C")  # (ascontiguousarray would promote a scalar to 1-d)
    if arr.dtype not in _DT_OF:
        raise ValueError("unsupported dtype %s" % arr.dtype)
    shape = b"".join(_enc_field(2, 2, _enc_field(1, 0, d)) for d in arr.shape)
    return (_enc_field(1, 0, _DT_OF[arr.dtype]) + _enc_field(2, 2, shape) +
            _enc_field(4, 2, arr.astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()))


def encode_const_node(name: str, arr: np.ndarray) -> bytes:
    arr = np.asarray(arr)
    attr_dtype = _enc_field(1, 2, b"dtype") + _enc_field(2, 2, _enc_field(6, 0, _DT_OF[arr.dtype]))
    attr_value_ = _enc_field(1, 2, b"value") + _enc_field(2, 2, _enc_field(8, 2, encode_tensor(arr)))
    return (_enc_field(1, 2, name.encode()) + _enc_field(2, 2, b"Const") + _enc_field(5, 2, attr_dtype) +
            _enc_field(5, 2, attr_value_))


def encode_op_node(name: str, op: str, inputs=(), dtype: Optional[int] = DT_FLOAT) -> bytes:
    body = _enc_field(1, 2, name.encode()) + _enc_field(2, 2, op.encode())
    for i in inputs:
        body += _enc_field(3, 2, i.encode())
    if dtype is not None:
        body += _enc_field(5, 2, _enc_field(1, 2, b"T") + _enc_field(2, 2, _enc_field(6, 0, dtype)))
    return body


def write_graph_consts(path: str, consts: Dict[str, np.ndarray], extra_nodes: Tuple[bytes, ...] = ()) -> None:
    """Serialize ``consts`` (+ optional pre-encoded op nodes) as a GraphDef readable by TensorFlow tools."""
    with open(path, "wb") as f:
        for name, arr in consts.items():
            f.write(_enc_field(1, 2, encode_const_node(name, arr)))
        for body in extra_nodes:
            f.write(_enc_field(1, 2, body))


def write_retrained_graph(path: str, final_weights: np.ndarray, final_biases: np.ndarray,
                          final_tensor_name: str = "final_result") -> None:
    """Frozen head in the reference's naming (``retrain1/retrain.py:262-279,470-473``):
    ``final_training_ops/weights/final_weights``, ``.../biases/final_biases`` and the
    ``MatMul -> add -> Softmax(final_tensor_name)`` chain reading ``pool_3/_reshape``."""
    w = "final_training_ops/weights/final_weights"
    b = "final_training_ops/biases/final_biases"
    nodes = (
        encode_op_node("pool_3/_reshape", "Placeholder", (), DT_FLOAT),
        encode_op_node("final_training_ops/Wx_plus_b/MatMul", "MatMul", ("pool_3/_reshape", w)),
        encode_op_node("final_training_ops/Wx_plus_b/add", "Add", ("final_training_ops/Wx_plus_b/MatMul", b)),
        encode_op_node(final_tensor_name, "Softmax", ("final_training_ops/Wx_plus_b/add",)),
    )
    write_graph_consts(path, {w: np.asarray(final_weights, np.float32), b: np.asarray(final_biases, np.float32)}, nodes)


# ----------------------------------------------------------------------------- the whole frozen network as a GraphDef
def _attr(key: str, value_payload: bytes) -> bytes:
    """One ``map<string, AttrValue>`` entry of a NodeDef (field 5)."""
    return _enc_field(5, 2, _enc_field(1, 2, key.encode()) + _enc_field(2, 2, value_payload))


def attr_type(dt: int) -> bytes: return _enc_field(6, 0, dt)
def attr_s(s: bytes) -> bytes: return _enc_field(2, 2, s)
def attr_b(b: bool) -> bytes: return _enc_field(5, 0, 1 if b else 0)
def attr_i(i: int) -> bytes: return _enc_field(3, 0, i)
def attr_f(f: float) -> bytes: return _enc_field(4, 5, struct.pack("<f", f))
def attr_ints(vals) -> bytes: return _enc_field(1, 2, _enc_field(3, 2, b"".join(_enc_varint(int(v)) for v in vals)))  # list { i: [...] }


def encode_node(name: str, op: str, inputs=(), attrs: Optional[Dict[str, bytes]] = None) -> bytes:
    body = _enc_field(1, 2, name.encode()) + _enc_field(2, 2, op.encode())
    for i in inputs:
        body += _enc_field(3, 2, i.encode())
    for k, v in (attrs or {}).items():
        body += _attr(k, v)
    return body


def attr_list_ints(raw) -> List[int]:
    """Decode an AttrValue holding ``list { i: ... }`` (strides / ksize)."""
    out: List[int] = []
    for fno, _wt, v in _fields(raw):
        if fno == 1:
            for f2, w2, v2 in _fields(v):
                if f2 == 3:
                    _packed_or_single(out, w2, v2, None, 0)
    return out


HEAD_W = "final_training_ops/weights/final_weights"
HEAD_B = "final_training_ops/biases/final_biases"
JPEG_DATA_TENSOR = "DecodeJpeg/contents"          # retrain1/retrain.py:34
RESIZED_INPUT_TENSOR = "ResizeBilinear"           # :35
BOTTLENECK_TENSOR = "pool_3/_reshape"             # :30


def write_inception_graphdef(path: str, graph, params, head_w: np.ndarray, head_b: np.ndarray,
                             final_tensor_name: str = "final_result") -> int:
    """``retrained_graph.pb`` as the reference writes it (``retrain1/retrain.py:470-473``:
    ``convert_variables_to_constants`` over the imported Inception graph + the new layer): ONE serialized GraphDef that
    holds the whole network under the 2015 graph's names [ext] —

    * ``DecodeJpeg/contents`` -> DecodeJpeg -> Cast -> ExpandDims -> ``ResizeBilinear`` (299x299) -> Sub 128 -> Mul 1/128;
    * per conv layer ``<L>/conv2d_params`` (HWIO), ``<L>/Conv2D``, ``<L>/batchnorm/{beta,gamma,moving_mean,
      moving_variance}``, ``<L>/batchnorm`` (BatchNormWithGlobalNormalization, eps 1e-3), ``<L>`` (Relu);
    * MaxPool / AvgPool nodes, ``<block>/join`` Concat nodes (channel axis 3), ``pool_3`` (8x8 AvgPool),
      ``pool_3/_reshape``;
    * ``final_training_ops/...`` MatMul + add and the Softmax named ``final_tensor_name``.

    ``graph``: ``models.inception_v3.build_graph()``; ``params``: ``{layer: {weights, beta, gamma, moving_mean,
    moving_variance}}`` (torch tensors or arrays).  Returns the number of nodes written."""
    F = attr_type(DT_FLOAT)
    nodes: List[bytes] = []

    def const(name: str, arr) -> str:
        nodes.append(encode_const_node(name, np.asarray(arr)))
        return name

    def node(name, op, inputs=(), attrs=None) -> str:
        nodes.append(encode_node(name, op, inputs, attrs))
        return name

    # ---- the input pipeline inside the graph (component C41)
    node(JPEG_DATA_TENSOR, "Placeholder", (), {"dtype": attr_type(DT_STRING)})
    node("DecodeJpeg", "DecodeJpeg", (JPEG_DATA_TENSOR,), {"channels": attr_i(3)})
    node("Cast", "Cast", ("DecodeJpeg",), {"SrcT": attr_type(DT_UINT8), "DstT": F})
    const("ExpandDims/dim", np.asarray(0, np.int32))
    node("ExpandDims", "ExpandDims", ("Cast", "ExpandDims/dim"), {"T": F})
    const("ResizeBilinear/size", np.asarray([299, 299], np.int32))
    node(RESIZED_INPUT_TENSOR, "ResizeBilinear", ("ExpandDims", "ResizeBilinear/size"), {"T": F, "align_corners": attr_b(False)})
    const("Sub/y", np.asarray(128.0, np.float32))
    node("Sub", "Sub", (RESIZED_INPUT_TENSOR, "Sub/y"), {"T": F})
    const("Mul/y", np.asarray(1.0 / 128.0, np.float32))
    node("Mul", "Mul", ("Sub", "Mul/y"), {"T": F})

    producers: Dict[str, List[Tuple[int, str]]] = {"input": [(0, "Mul")]}
    joined: Dict[str, str] = {}

    def tensor_of(buf: str) -> str:
        """The node that holds buffer ``buf``: its single producer, or a Concat over the producers' channel slices."""
        ps = sorted(producers[buf])
        if len(ps) == 1:
            return ps[0][1]
        if buf not in joined:
            const(buf + "/join/concat_dim", np.asarray(3, np.int32))
            joined[buf] = node(buf + "/join", "Concat", [buf + "/join/concat_dim"] + [n for _o, n in ps],
                               {"T": F, "N": attr_i(len(ps))})
        return joined[buf]

    def pool_attrs(k: int, stride: int, padding: bytes) -> Dict[str, bytes]:
        return {"T": F, "ksize": attr_ints([1, k, k, 1]), "strides": attr_ints([1, stride, stride, 1]), "padding": attr_s(padding)}

    for op in graph.ops:
        src = tensor_of(op.src)
        if op.kind == "conv":
            s = op.conv
            p = params[s.name]
            as_np = lambda t: np.ascontiguousarray(np.asarray(t.detach().cpu() if hasattr(t, "detach") else t, np.float32))
            w = const(s.name + "/conv2d_params", as_np(p["weights"]))
            valid = s.pad == (0, 0) and (s.kh > 1 or s.kw > 1)
            conv = node(s.name + "/Conv2D", "Conv2D", (src, w),
                        {"T": F, "strides": attr_ints([1, s.stride, s.stride, 1]), "padding": attr_s(b"VALID" if valid else b"SAME"),
                         "use_cudnn_on_gpu": attr_b(True)})
            bn = [const("%s/batchnorm/%s" % (s.name, k), as_np(p[k])) for k in ("moving_mean", "moving_variance", "beta", "gamma")]
            node(s.name + "/batchnorm", "BatchNormWithGlobalNormalization", [conv] + bn,
                 {"T": F, "variance_epsilon": attr_f(1e-3), "scale_after_normalization": attr_b(False)})
            out = node(s.name, "Relu", (s.name + "/batchnorm",), {"T": F})
        elif op.kind == "maxpool":
            name = op.dst[:-4] if op.dst.endswith(":out") else op.dst + "/pool"
            out = node(name, "MaxPool", (src,), pool_attrs(3, op.stride, b"VALID"))
        elif op.kind == "avgpool":
            out = node(op.dst[:-4], "AvgPool", (src,), pool_attrs(3, 1, b"SAME"))
        elif op.kind == "gap":
            out = node("pool_3", "AvgPool", (src,), pool_attrs(8, 1, b"VALID"))
        else:
            raise ValueError("unknown op kind %r" % op.kind)
        producers.setdefault(op.dst, []).append((op.dst_off, out))
    const(BOTTLENECK_TENSOR + "/shape", np.asarray([1, 2048], np.int32))
    node(BOTTLENECK_TENSOR, "Reshape", ("pool_3", BOTTLENECK_TENSOR + "/shape"), {"T": F})
    # ---- the new layer (retrain1/retrain.py:262-279)
    const(HEAD_W, np.asarray(head_w, np.float32))
    const(HEAD_B, np.asarray(head_b, np.float32))
    node("final_training_ops/Wx_plus_b/MatMul", "MatMul", (BOTTLENECK_TENSOR, HEAD_W),
         {"T": F, "transpose_a": attr_b(False), "transpose_b": attr_b(False)})
    node("final_training_ops/Wx_plus_b/add", "Add", ("final_training_ops/Wx_plus_b/MatMul", HEAD_B), {"T": F})
    node(final_tensor_name, "Softmax", ("final_training_ops/Wx_plus_b/add",), {"T": F})
    with open(path, "wb") as f:
        for body in nodes:
            f.write(_enc_field(1, 2, body))
    return len(nodes)


def is_graphdef(path: str) -> bool:
    """A serialized GraphDef starts with a NodeDef field (tag 0x0a); the native checkpoint container with its magic."""
    with open(path, "rb") as f:
        head = f.read(8)
    return len(head) > 0 and head[0] == 0x0A and not head.startswith(b"DTFCKPT")


def read_retrained_graph(path: str):
    """What ``retrain1/test.py:26-39`` gets from ``retrained_graph.pb``: the backbone's per-layer weights, the new layer and
    the name of the output tensor (the Softmax that reads the new layer's ``add``).  Also checks that the file is the network
    it claims to be: the reference's tensor names exist and every conv layer has its five constants.
    Returns (layers, final_weights, final_biases, final_tensor_name, ops)."""
    consts: Dict[str, np.ndarray] = {}
    ops: List[Tuple[str, str, List[str]]] = []
    for n in iter_nodes(path):
        ops.append((n["name"], n["op"], n["input"]))
        if n["op"] == "Const" and "value" in n["attr"] and attr_value(n["attr"].get("dtype")) == DT_FLOAT:
            consts[n["name"]] = np.array(attr_value(n["attr"]["value"]))
    names = {n for n, _o, _i in ops}
    for need in (JPEG_DATA_TENSOR, RESIZED_INPUT_TENSOR, BOTTLENECK_TENSOR, HEAD_W, HEAD_B):
        if need not in names:
            raise ValueError("%s: no node named %r (not a retrained Inception-v3 graph)" % (path, need))
    final = [n for n, o, i in ops if o == "Softmax" and i and i[0].startswith("final_training_ops/")]
    if len(final) != 1:
        raise ValueError("%s: expected one Softmax over the new layer, found %s" % (path, final))
    layers: Dict[str, Dict[str, np.ndarray]] = {}
    for name, arr in consts.items():
        if name.endswith("/conv2d_params"):
            layers.setdefault(name[:-len("/conv2d_params")], {})["weights"] = arr
        else:
            for k in _BN_KEYS:
                if name.endswith("/batchnorm/" + k):
                    layers.setdefault(name[:-len("/batchnorm/" + k)], {})[k] = arr
    bad = [k for k, v in layers.items() if set(v) != {"weights"} | set(_BN_KEYS)]
    if bad:
        raise ValueError("%s: incomplete conv layers %s" % (path, bad[:4]))
    return layers, consts[HEAD_W], consts[HEAD_B], final[0], ops
