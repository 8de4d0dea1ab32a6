"""TF-free GraphDef reader/writer (C26/C28): round trip + cross-check against google.protobuf with a
dynamically declared subset of tensorflow's graph.proto (TensorFlow itself is not installed)."""
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
