"""TF-free GraphDef reader/writer (C26/C28): round trip + cross-check against google.protobuf with a
dynamically declared subset of tensorflow's graph.proto (TensorFlow itself is not installed)."""
import os
import numpy as np
import pytest

from distributed_tensorflow_amd.utils import graphdef as gd


def _tf_subset_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fdp = descriptor_pb2.FileDescriptorProto(name="dtf_test_graph.proto", package="dtftest", syntax="proto3")

    def msg(name, fields):
        m = fdp.message_type.add(name=name)
        for fname, num, ftype, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".dtftest." + tname
        return m
    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("Dim", [("size", 1, F.TYPE_INT64, OPT, None)])
    msg("TensorShapeProto", [("dim", 2, F.TYPE_MESSAGE, REP, "Dim")])
    msg("TensorProto", [("dtype", 1, F.TYPE_INT32, OPT, None), ("tensor_shape", 2, F.TYPE_MESSAGE, OPT, "TensorShapeProto"),
                        ("tensor_content", 4, F.TYPE_BYTES, OPT, None), ("float_val", 5, F.TYPE_FLOAT, REP, None),
                        ("int_val", 7, F.TYPE_INT32, REP, None)])
    msg("AttrValue", [("s", 2, F.TYPE_BYTES, OPT, None), ("i", 3, F.TYPE_INT64, OPT, None), ("f", 4, F.TYPE_FLOAT, OPT, None),
                      ("type", 6, F.TYPE_INT32, OPT, None), ("tensor", 8, F.TYPE_MESSAGE, OPT, "TensorProto")])
    msg("AttrEntry", [("key", 1, F.TYPE_STRING, OPT, None), ("value", 2, F.TYPE_MESSAGE, OPT, "AttrValue")])
    msg("NodeDef", [("name", 1, F.TYPE_STRING, OPT, None), ("op", 2, F.TYPE_STRING, OPT, None),
                    ("input", 3, F.TYPE_STRING, REP, None), ("attr", 5, F.TYPE_MESSAGE, REP, "AttrEntry")])
    msg("GraphDef", [("node", 1, F.TYPE_MESSAGE, REP, "NodeDef")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("dtftest." + n))
    return get("GraphDef"), get("TensorProto")


def test_round_trip_consts(tmp_path):
    rng = np.random.RandomState(0)
    consts = {"conv/conv2d_params": rng.randn(3, 3, 3, 32).astype(np.float32),
              "conv/batchnorm/beta": rng.randn(32).astype(np.float32),
              "some/int": np.arange(6, dtype=np.int32).reshape(2, 3),
              "scalar": np.float32(2.5)}
    p = str(tmp_path / "g.pb")
    gd.write_graph_consts(p, consts)
    back = gd.read_graph_consts(p, dtypes=None)
    assert set(back) == set(consts)
    for k in consts:
        assert back[k].dtype == np.asarray(consts[k]).dtype and np.array_equal(back[k], consts[k])
    assert set(gd.read_graph_consts(p)) == {"conv/conv2d_params", "conv/batchnorm/beta", "scalar"}


def test_writer_is_valid_protobuf_and_reader_takes_protobuf_output(tmp_path):
    GraphDef, TensorProto = _tf_subset_classes()
    w = np.random.RandomState(1).randn(8, 5).astype(np.float32)
    b = np.zeros(5, np.float32)
    p = str(tmp_path / "retrained_graph.pb")
    gd.write_retrained_graph(p, w, b, "final_result")
    g = GraphDef()
    g.ParseFromString(open(p, "rb").read())
    names = [n.name for n in g.node]
    assert names[:2] == ["final_training_ops/weights/final_weights", "final_training_ops/biases/final_biases"]
    assert names[-1] == "final_result" and g.node[-1].op == "Softmax"
    t = [a.value.tensor for a in g.node[0].attr if a.key == "value"][0]
    assert t.dtype == gd.DT_FLOAT and [d.size for d in t.tensor_shape.dim] == [8, 5]
    assert np.array_equal(np.frombuffer(t.tensor_content, np.float32).reshape(8, 5), w)
    # the other direction: a graph serialized by google.protobuf, using float_val / splat / int_val encodings
    g2 = GraphDef()
    n = g2.node.add(name="a/conv2d_params", op="Const")
    e = n.attr.add(key="dtype"); e.value.type = gd.DT_FLOAT
    e = n.attr.add(key="value"); e.value.tensor.dtype = gd.DT_FLOAT
    for d in (1, 1, 2, 3):
        e.value.tensor.tensor_shape.dim.add(size=d)
    e.value.tensor.float_val.extend([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    n = g2.node.add(name="a/batchnorm/beta", op="Const")
    e = n.attr.add(key="dtype"); e.value.type = gd.DT_FLOAT
    e = n.attr.add(key="value"); e.value.tensor.dtype = gd.DT_FLOAT
    e.value.tensor.tensor_shape.dim.add(size=3)
    e.value.tensor.float_val.append(0.5)  # splat
    n = g2.node.add(name="mul", op="Mul"); n.input.extend(["a/conv2d_params", "a/batchnorm/beta"])
    p2 = str(tmp_path / "g2.pb")
    open(p2, "wb").write(g2.SerializeToString())
    layers = gd.inception_weights_from_graphdef(p2)
    assert list(layers) == ["a"]
    assert layers["a"]["weights"].shape == (1, 1, 2, 3) and layers["a"]["weights"].ravel().tolist() == [1, 2, 3, 4, 5, 6]
    assert layers["a"]["beta"].tolist() == [0.5, 0.5, 0.5]
    assert gd.read_graph_ops(p2)[-1] == ("mul", "Mul", ["a/conv2d_params", "a/batchnorm/beta"])


def test_inception_weights_load_into_model(tmp_path):
    import torch
    from distributed_tensorflow_amd.models.inception_v3 import InceptionV3
    m = InceptionV3(device="cpu", seed=0)
    rng = np.random.RandomState(2)
    consts = {}
    for name in ("conv", "mixed_10/tower_1/mixed/conv_1"):
        shp = tuple(m.params[name]["weights"].shape)
        consts[name + "/conv2d_params"] = rng.randn(*shp).astype(np.float32)
        for k in ("beta", "gamma", "moving_mean"):
            consts[name + "/batchnorm/" + k] = rng.randn(shp[-1]).astype(np.float32)
        consts[name + "/batchnorm/moving_variance"] = rng.rand(shp[-1]).astype(np.float32) + 0.5
    p = str(tmp_path / "classify_image_graph_def.pb")
    gd.write_graph_consts(p, consts)
    m.load_tf_weights(gd.inception_weights_from_graphdef(p))
    assert torch.equal(m.params["conv"]["weights"], torch.from_numpy(consts["conv/conv2d_params"]))
    assert torch.equal(m.params["mixed_10/tower_1/mixed/conv_1"]["moving_variance"],
                       torch.from_numpy(consts["mixed_10/tower_1/mixed/conv_1/batchnorm/moving_variance"]))
    with pytest.raises(KeyError):
        m.load_tf_weights({"nope": {"weights": np.zeros((1, 1, 1, 1), np.float32)}})


def test_whole_network_graphdef_round_trip(tmp_path):
    """``retrained_graph.pb`` is a real GraphDef of Inception-v3 + the new layer (``/root/reference/retrain1/retrain.py:470-473``)
    under the tensor names the reference's scripts use (``retrain.py:29-35``, ``test.py:33,39``): export -> parse without
    TensorFlow -> a fresh network holds bit-identical parameters; the topology is the 2015 graph's (94 Conv2D, 11 joins)."""
    import torch
    from distributed_tensorflow_amd.models.inception_v3 import InceptionV3
    from distributed_tensorflow_amd.utils import graphdef as gd
    net = InceptionV3(device="cpu", seed=11)
    g = torch.Generator().manual_seed(4)
    for p in net.params.values():   # non-trivial batch-norm constants
        c = p["beta"].shape[0]
        p["beta"], p["gamma"] = torch.randn(c, generator=g) * 0.1, 1.0 + 0.1 * torch.randn(c, generator=g)
        p["moving_mean"], p["moving_variance"] = torch.randn(c, generator=g) * 0.1, 0.5 + torch.rand(c, generator=g)
    fw, fb = np.random.RandomState(0).randn(2048, 5).astype(np.float32), np.arange(5, dtype=np.float32)
    path = str(tmp_path / "retrained_graph.pb")
    n = gd.write_inception_graphdef(path, net.graph, net.params, fw, fb, "final_result")
    assert gd.is_graphdef(path) and os.path.getsize(path) > 80e6
    layers, fw2, fb2, final_name, ops = gd.read_retrained_graph(path)
    assert final_name == "final_result" and np.array_equal(fw, fw2) and np.array_equal(fb, fb2) and len(ops) == n
    by_op = {}
    for _name, op, _inp in ops:
        by_op[op] = by_op.get(op, 0) + 1
    assert by_op["Conv2D"] == 94 and by_op["BatchNormWithGlobalNormalization"] == 94 and by_op["Relu"] == 94
    assert by_op["Concat"] == 11 and by_op["MaxPool"] == 4 and by_op["AvgPool"] == 10 and by_op["Softmax"] == 1
    names = {n_: (op, inp) for n_, op, inp in ops}
    assert names["DecodeJpeg"][1] == ["DecodeJpeg/contents"] and names["ResizeBilinear"][0] == "ResizeBilinear"
    assert names["conv/Conv2D"][1] == ["Mul", "conv/conv2d_params"]
    assert names["pool_3/_reshape"] == ("Reshape", ["pool_3", "pool_3/_reshape/shape"])
    assert names["mixed/join"][1] == ["mixed/join/concat_dim", "mixed/conv", "mixed/tower/conv_1", "mixed/tower_1/conv_2",
                                      "mixed/tower_2/conv"]
    assert names["mixed_3/join"][1][1:] == ["mixed_3/conv", "mixed_3/tower/conv_2", "mixed_3/pool"]
    assert names["mixed_10/tower_2/pool"] == ("AvgPool", ["mixed_9/join"]) and names["pool_3"][1] == ["mixed_10/join"]
    # attributes survive: strides / padding of a strided VALID conv and of a SAME 1x7
    raw = {n_["name"]: n_ for n_ in gd.iter_nodes(path) if n_["name"] in ("conv/Conv2D", "mixed_4/tower/conv_1/Conv2D", "pool_3")}
    assert gd.attr_list_ints(raw["conv/Conv2D"]["attr"]["strides"]) == [1, 2, 2, 1]
    assert gd.attr_value(raw["conv/Conv2D"]["attr"]["padding"]) == b"VALID"
    assert gd.attr_value(raw["mixed_4/tower/conv_1/Conv2D"]["attr"]["padding"]) == b"SAME"
    assert gd.attr_list_ints(raw["pool_3"]["attr"]["ksize"]) == [1, 8, 8, 1]
    # a fresh network loaded from the file is the exporting network, bit for bit
    fresh = InceptionV3(device="cpu", seed=999)
    fresh.load_tf_weights(layers)
    for name in net.graph.convs:
        for k in ("weights", "beta", "gamma", "moving_mean", "moving_variance"):
            assert torch.equal(fresh.params[name][k], net.params[name][k].float()), (name, k)
        assert torch.equal(fresh.folded(name)[0], net.folded(name)[0])
    # the older importer entry point reads the same file (the first fixture it sees that another function wrote)
    assert set(gd.inception_weights_from_graphdef(path)) == set(net.graph.convs)
