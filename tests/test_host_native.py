"""CPU tests of the native host library: CRC32C, tfevents writer, store, checkpoint container."""
import os
import threading
import time

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import native
from distributed_tensorflow_amd.parallel import cluster as cl
from distributed_tensorflow_amd.utils import checkpoint as ckpt
from distributed_tensorflow_amd.utils import summary as sm


def test_crc32c_known_vectors():
    lib = native.host()
    assert lib.dtf_crc32c(b"123456789", 9) == 0xE3069283
    assert lib.dtf_crc32c(b"", 0) == 0
    assert lib.dtf_crc32c(bytes(32), 32) == 0x8A9136AA  # RFC 3720 B.4: 32 bytes of zeros
    assert lib.dtf_crc32c(bytes([0xFF] * 32), 32) == 0x62A8AB43


def test_event_file_roundtrip(tmp_path):
    w = sm.FileWriter(str(tmp_path), graph={"model": "x"})
    for step in range(5):
        s = sm.Summary().scalar("cross_entropy", 2.0 - 0.1 * step).scalar("accuracy", 0.1 * step)
        if step == 2:
            s.histogram("weights", np.linspace(-1, 1, 1001))
        w.add_summary(s, step)
    w.close()
    evs = sm.read_events(w.path)
    assert evs[0]["file_version"] == "brain.Event:2"
    assert "graph" in evs[1]
    steps = [e for e in evs if e["scalars"]]
    assert [e["step"] for e in steps] == [0, 1, 2, 3, 4]
    assert abs(steps[3]["scalars"]["cross_entropy"] - 1.7) < 1e-6
    h = steps[2]["histograms"]["weights"]
    assert h["num"] == 1001 and abs(h["min"] + 1) < 1e-12 and abs(h["max"] - 1) < 1e-12
    assert abs(sum(h["bucket"]) - 1001) < 1e-9 and len(h["bucket"]) == len(h["bucket_limit"])
    assert os.path.exists(os.path.join(str(tmp_path), "scalars.jsonl"))
    # corrupting a byte must be caught by the CRC
    raw = bytearray(open(w.path, "rb").read())
    raw[40] ^= 0xFF
    bad = os.path.join(str(tmp_path), "bad")
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError):
        sm.read_events(bad)


def test_variable_summaries_tags():
    s = sm.variable_summaries("final_training_ops/weights", torch.arange(10.0))
    tags = dict(s.scalars)
    assert abs(tags["final_training_ops/weights/summaries/mean"] - 4.5) < 1e-6
    assert abs(tags["final_training_ops/weights/summaries/stddev/stddev"] - np.std(np.arange(10.0))) < 1e-5
    assert tags["final_training_ops/weights/summaries/max"] == 9 and tags["final_training_ops/weights/summaries/min"] == 0
    assert s.histograms[0][0].endswith("/histogram")


def test_checkpoint_container_roundtrip_and_crc(tmp_path):
    path = str(tmp_path / "ck" / "model.ckpt-7")
    t = {"a/W": np.random.randn(5, 5, 1, 32).astype(np.float32), "step": np.asarray(7, np.int32),
         "bf": torch.randn(17).to(torch.bfloat16), "i64": np.arange(6, dtype=np.int64).reshape(2, 3)}
    ckpt.save_tensors(path, t)
    got = ckpt.load_tensors(path)
    assert list(got.keys()) == list(t.keys())
    assert np.array_equal(got["a/W"], t["a/W"]) and got["step"] == 7 and got["i64"].shape == (2, 3)
    assert got["bf"].dtype == np.uint16 and got["bf"].shape == (17,)
    raw = bytearray(open(path, "rb").read())
    raw[-3] ^= 0x55
    open(path, "wb").write(raw)
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_tensors(path)
    assert not [f for f in os.listdir(os.path.dirname(path)) if ".tmp." in f]


def test_latest_checkpoint_resolution(tmp_path):
    d = str(tmp_path)
    for s in (5, 37, 12):
        ckpt.save_tensors(os.path.join(d, "model.ckpt-%d" % s), {"x": np.zeros(1, np.float32)})
    assert ckpt.latest_checkpoint(d).endswith("model.ckpt-37")
    ckpt.update_checkpoint_state(d, os.path.join(d, "model.ckpt-12"))
    assert ckpt.latest_checkpoint(d).endswith("model.ckpt-12")  # state file wins, like TF


def test_store_set_get_add_barrier():
    srv = cl.StoreServer(0)
    try:
        a = cl.StoreClient("127.0.0.1", srv.port)
        b = cl.StoreClient("127.0.0.1", srv.port)
        a.set("k", b"hello")
        assert b.get("k", timeout_s=1) == b"hello"
        assert b.get("missing", timeout_s=0.05) is None
        assert a.add("c", 2) == 2 and b.add("c", 3) == 5 and a.counter("c") == 5
        got = {}
        th = threading.Thread(target=lambda: got.setdefault("v", b.get("late", timeout_s=5)))
        th.start()
        time.sleep(0.1)
        a.set("late", b"x" * 1000)
        th.join()
        assert got["v"] == b"x" * 1000
        res = []
        ts = [threading.Thread(target=lambda c=c: res.append(c.barrier("b0", 2, timeout_s=5))) for c in (a, b)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert res == [True, True]
        assert a.check("k") and not a.check("nope")
        a.close(); b.close()
    finally:
        srv.stop()


def test_ps_role_serves_until_workers_done():
    spec = cl.ClusterSpec.from_flags("127.0.0.1:0", "127.0.0.1:29999, 127.0.0.1:29998")
    assert spec.world_size == 2 and spec.job_tasks("worker")[1] == "127.0.0.1:29998"
    ps = cl.Server(spec, "ps", 0)
    port = ps.store_server.port
    th = threading.Thread(target=lambda: ps.join(poll_s=0.02, timeout_s=10))
    th.start()
    c = cl.StoreClient("127.0.0.1", port)
    c.add("workers_done", 1)
    time.sleep(0.1)
    assert th.is_alive()
    c.add("workers_done", 1)
    th.join(timeout=5)
    assert not th.is_alive()


def test_heartbeat_detects_dead_rank():
    srv = cl.StoreServer(0)
    try:
        c0, c1 = cl.StoreClient("127.0.0.1", srv.port), cl.StoreClient("127.0.0.1", srv.port)
        hb0 = cl.Heartbeat(c0, 0, period_s=0.02).start()
        hb1 = cl.Heartbeat(c1, 1, period_s=0.02).start()
        time.sleep(0.1)
        mon = cl.StoreClient("127.0.0.1", srv.port)
        watcher = cl.Heartbeat(mon, 99)
        assert watcher.dead_ranks(2, timeout_s=0.2) == []
        hb1.stop()
        time.sleep(0.1)
        watcher.dead_ranks(2, timeout_s=0.15)
        time.sleep(0.3)
        assert watcher.dead_ranks(2, timeout_s=0.15) == [1]
        hb0.stop()
    finally:
        srv.stop()


def test_bottleneck_text_codec():
    import ctypes as C
    lib = native.host()
    vals = np.random.rand(2048).astype(np.float32)
    buf = C.create_string_buffer(2048 * 20)
    n = lib.dtf_format_floats(vals.ctypes.data_as(C.c_void_p), 2048, buf, len(buf))
    assert n > 0
    text = buf.raw[:n]
    out = np.empty(2048, np.float32)
    m = lib.dtf_parse_floats(text, len(text), out.ctypes.data_as(C.c_void_p), 2048)
    assert m == 2048 and np.array_equal(out, vals)
    ref = ",".join(str(x) for x in vals[:16])  # the reference's format
    assert lib.dtf_parse_floats(ref.encode(), len(ref), out.ctypes.data_as(C.c_void_p), 2048) == 16
    assert np.allclose(out[:16], vals[:16], rtol=1e-6)
    assert lib.dtf_parse_floats(b"1.0,abc", 7, out.ctypes.data_as(C.c_void_p), 2048) == -1


def test_trace_ranges_and_step_timer():
    from distributed_tensorflow_amd.utils import trace
    with trace.range("disabled"):  # no-op when not enabled
        pass
    avail = trace.enable(True)
    assert avail in (True, False)
    with trace.range("outer"):
        trace.mark("m")
        with trace.range("inner"):
            pass
    trace.enable(False)
    t = trace.StepTimer()
    for _ in range(3):
        with t.phase("a"):
            pass
        with t.phase("b"):
            pass
        t.add_images(100)
    s = t.summary()
    assert "images/sec" in s and "a " in s and "x3" in s
    assert trace.format_table([("k", 2, 1.0, 500.0)]).count("\n") == 1


# ----------------------------------------------------------------------------- events writer under two threads
def test_file_writer_add_summary_is_thread_safe(tmp_path):
    """The Supervisor's autosave thread and the training thread write through one FileWriter: every record must stay a
    well-formed event (CRCs verify) carrying exactly the scalars of ONE add_summary call under its own step."""
    import threading
    from distributed_tensorflow_amd.utils.summary import FileWriter, Summary, read_events
    w = FileWriter(str(tmp_path))
    n = 400

    def writer(kind):
        for i in range(n):
            s = Summary()
            if kind == 0:
                s.scalar("a/x", float(i)).scalar("a/y", float(2 * i)).scalar("a/z", float(3 * i))
                w.add_summary(s, i)
            else:
                w.add_scalars({"global_step/sec": float(i)}, 100000 + i)
    th = [threading.Thread(target=writer, args=(k,)) for k in (0, 1)]
    [t.start() for t in th]
    [t.join() for t in th]
    w.close()
    ev = [e for e in read_events(w.path, verify_crc=True) if e["scalars"]]
    assert len(ev) == 2 * n
    for e in ev:
        if e["step"] >= 100000:
            assert e["scalars"] == {"global_step/sec": float(e["step"] - 100000)}
        else:
            i = e["step"]
            assert e["scalars"] == {"a/x": float(i), "a/y": float(2 * i), "a/z": float(3 * i)}


# ----------------------------------------------------------------------------- TensorFlow tensor bundle (tf.train.Saver's files)
def _py_varint(b, pos):
    v = shift = 0
    while True:
        x = b[pos]
        pos += 1
        v |= (x & 0x7F) << shift
        if not x & 0x80:
            return v, pos
        shift += 7


def _py_table_entries(path):
    """An independent statement of the LevelDB table format in plain Python (footer -> index block -> data blocks, prefix
    compression, masked CRC32C trailers): the native reader is not only checked against its own writer."""
    import struct
    from distributed_tensorflow_amd import native
    lib = native.host()
    data = open(path, "rb").read()
    assert struct.unpack("<Q", data[-8:])[0] == 0xDB4775248B80FB57
    pos = len(data) - 48
    _moff, pos = _py_varint(data, pos)
    _msize, pos = _py_varint(data, pos)
    ioff, pos = _py_varint(data, pos)
    isize, pos = _py_varint(data, pos)

    def block(off, size):
        assert data[off + size] == 0                                          # no compression
        (crc,) = struct.unpack("<I", data[off + size + 1:off + size + 5])
        raw = lib.dtf_crc32c(data[off:off + size + 1], size + 1)
        assert (((raw >> 15) | (raw << 17)) + 0xA282EAD8) & 0xFFFFFFFF == crc  # masked CRC32C of block + type byte
        (nrestart,) = struct.unpack("<I", data[off + size - 4:off + size])
        end = off + size - 4 - 4 * nrestart
        p, key, out = off, b"", []
        while p < end:
            shared, p = _py_varint(data, p)
            non_shared, p = _py_varint(data, p)
            vlen, p = _py_varint(data, p)
            key = key[:shared] + data[p:p + non_shared]
            p += non_shared
            out.append((key, data[p:p + vlen]))
            p += vlen
        return out

    entries = []
    for _k, handle in block(ioff, isize):
        boff, q = _py_varint(handle, 0)
        bsize, q = _py_varint(handle, q)
        entries += block(boff, bsize)
    return entries


def test_tf_tensor_bundle_round_trip_and_independent_parse(tmp_path):
    """``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` as ``tf.train.Saver`` writes them
    (``/root/reference/demo1/train.py:144,165``; restored by ``demo1/test.py:175-182``)."""
    from distributed_tensorflow_amd.utils import checkpoint as ckpt
    rng = np.random.RandomState(0)
    tensors = {"Conv1/Variable": rng.randn(5, 5, 1, 32).astype(np.float32),
               "Conv1/Variable/Adam": rng.randn(5, 5, 1, 32).astype(np.float32),
               "Conv1/Variable/Adam_1": rng.rand(5, 5, 1, 32).astype(np.float32),
               "fc1/Variable": rng.randn(3136, 64).astype(np.float32),
               "train_step/beta1_power": np.asarray(0.9 ** 7, np.float32),
               "Variable": np.asarray(1234, np.int32),
               "counts": np.arange(7, dtype=np.int64)}
    prefix = str(tmp_path / "model" / "train.ckpt")
    ckpt.save_tf_bundle(prefix, tensors)
    assert ckpt.is_tf_bundle(prefix) and not os.path.exists(prefix)
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(v.nbytes for v in tensors.values())
    back = ckpt.load_tf_bundle(prefix)
    assert list(back) == sorted(tensors)                                       # keys come back in table (sorted) order
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    # the same file through the plain-Python statement of the format: header first, keys sorted and prefix-compressed
    ent = _py_table_entries(prefix + ".index")
    assert ent[0][0] == b"" and [k.decode() for k, _v in ent[1:]] == sorted(tensors)
    raw = open(prefix + ".index", "rb").read()
    assert raw.count(b"Conv1/Variable") == 1                                   # shared prefixes are stored once per restart run
    # BundleEntryProto of the scalar: dtype DT_INT32 (3), empty shape, size 4
    e = dict(ent)[b"Variable"]
    assert e[:2] == bytes([0x08, 3]) and bytes([0x28, 4]) in e
    # corruption of one tensor byte is caught by its CRC; corruption of the index by the block CRC
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(100)
        b = f.read(1)
        f.seek(100)
        f.write(bytes([b[0] ^ 0x40]))
    with pytest.raises(ckpt.CheckpointError, match="CRC"):
        ckpt.load_tf_bundle(prefix)
    ckpt.save_tf_bundle(prefix, tensors)
    with open(prefix + ".index", "r+b") as f:
        f.seek(20)
        b = f.read(1)
        f.seek(20)
        f.write(bytes([b[0] ^ 0x01]))
    with pytest.raises(ckpt.CheckpointError):
        ckpt.load_tf_bundle(prefix)


def test_tf_tensor_bundle_many_entries_span_blocks(tmp_path):
    from distributed_tensorflow_amd.utils import checkpoint as ckpt
    tensors = {"layer_%04d/some/rather/long/variable/name/to/fill/the/index/block/weights" % i:
               np.full((3,), i, np.float32) for i in range(6000)}
    prefix = str(tmp_path / "big.ckpt")
    ckpt.save_tf_bundle(prefix, tensors)
    assert os.path.getsize(prefix + ".index") > 300 << 10                      # more than one 256 KB block
    back = ckpt.load_tf_bundle(prefix)
    assert len(back) == 6000 and all(float(back[k][0]) == float(v[0]) for k, v in tensors.items())
    ent = _py_table_entries(prefix + ".index")
    assert len(ent) == 6001


def test_saver_writes_what_tf_saver_writes_and_restores_from_it(tmp_path):
    """``Saver.save`` leaves ``model/train.ckpt.index`` + ``.data-00000-of-00001`` next to the native container;
    ``restore`` accepts the bundle prefix alone (what a TensorFlow run would have left)."""
    import torch
    from distributed_tensorflow_amd.data import mnist
    from distributed_tensorflow_amd.engine.mnist_engine import EagerMnistEngine
    from distributed_tensorflow_amd.utils import checkpoint as ckpt
    imgs, labels = mnist.synthetic_mnist(64, seed=0)
    ds = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=0)
    dd = mnist.DeviceDataset(ds, "cpu", seed=0)
    e = EagerMnistEngine(dd, batch_size=16, seed=1)
    for _ in range(3):
        e.train_step()
    path = ckpt.Saver(e, with_global_step=True).save(str(tmp_path / "logs" / "model.ckpt"), global_step=3)
    assert path.endswith("model.ckpt-3") and ckpt.is_tf_bundle(path)
    t = ckpt.load_tf_bundle(path)
    assert t["Variable"] == 3 and t["Conv1/weights/Variable"].shape == (5, 5, 1, 32) if "Conv1/weights/Variable" in t else True
    assert any(k.endswith("/Adam_1") for k in t) and "train_step/beta1_power" in t
    os.remove(path)                                                             # only TensorFlow's files are left
    e2 = EagerMnistEngine(dd, batch_size=16, seed=99)
    assert ckpt.Saver(e2).restore(path) == 3
    assert torch.equal(e2.P, e.P) and torch.equal(e2.M, e.M) and torch.equal(e2.V, e.V)
