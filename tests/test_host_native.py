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
