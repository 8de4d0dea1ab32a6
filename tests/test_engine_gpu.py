"""End-to-end checks of the native MNIST step executor against the PyTorch oracle (GPU only)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dataset(n=2000, seed=0):
    from distributed_tensorflow_amd.data import mnist
    imgs, labels = mnist.synthetic_mnist(n, seed=seed)
    ds = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=seed)
    return mnist.DeviceDataset(ds, "cuda", seed=seed)


def _relerr(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


@pytest.mark.parametrize("double_softmax,B", [(True, 100), (False, 100), (True, 1), (True, 37), (False, 128),
                                              (True, 200), (False, 333)])
def test_step_gradients_match_oracle(double_softmax, B):
    """One step of the native executor vs autograd on the same batch / dropout mask, for the reference's batch
    (100), odd batches, and batches beyond 128 (separate fc dW launches instead of the grouped one)."""
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    from distributed_tensorflow_amd.models import mnist_cnn
    dd = _dataset()
    eng = NativeMnistEngine(dd, batch_size=B, keep_prob=0.7, double_softmax=double_softmax, seed=3,
                            use_graph=False, eval_chunk=500)
    params = {k: v.clone() for k, v in eng.param_views().items()}
    eng.fwd_bwd(3)
    eng.stream.synchronize()
    idx = dd.batch_indices(0, B)
    x, y = dd.gather(idx)
    h_drop = eng.ws["h_drop"].view(-1, 1024)[:B].float()
    mask = (h_drop != 0).float()
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    # the executor's own ReLU-and-dropout gate: see forward_logits(fc1_gate=...)
    logits = mnist_cnn.forward_logits(leaf, x, keep_prob=0.7, fc1_gate=mask, training=True, bf16_sim=True)
    loss, pred = mnist_cnn.softmax_xent(logits, y.long(), double_softmax=double_softmax)
    grads = torch.autograd.grad(loss, list(leaf.values()))
    g_native = eng.param_views(eng.G)
    # dropout keep-rate sanity: kept fraction among positive activations ~ keep_prob
    logits_nodrop, inter = mnist_cnn.forward_logits(params, x, bf16_sim=True, return_intermediates=True)
    pos = inter["h_drop"] > 0
    kept = float(((h_drop != 0) & pos).sum()) / float(pos.sum())
    assert abs(kept - 0.7) < (0.02 if B >= 37 else 0.1)
    lm, acc = eng.read_metrics(0, 1)
    assert abs(float(lm[0]) - float(loss.detach())) < 2e-3 * max(1.0, abs(float(loss.detach())))
    assert abs(float(acc[0]) - float((pred.argmax(1) == y.long()).float().mean())) < 1e-6
    for (name, g_ref) in zip(leaf.keys(), grads):
        err = _relerr(g_native[name], g_ref)
        assert err < 3e-2, (name, err)


def test_graph_replay_matches_direct_and_learns():
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    dd1, dd2 = _dataset(seed=1), _dataset(seed=1)
    e1 = NativeMnistEngine(dd1, batch_size=100, seed=5, use_graph=True, eval_chunk=500, lr=1e-4)
    e2 = NativeMnistEngine(dd2, batch_size=100, seed=5, use_graph=False, eval_chunk=500, lr=1e-4)
    e1.train_step()
    e2.train_step()
    e1.stream.synchronize(); e2.stream.synchronize()
    # one step: identical kernels, only fp32 atomic ordering differs (Adam's first update is +-lr
    # for every element, so compare the gradient-independent magnitude loosely and the metrics tightly)
    assert float((e1.P - e2.P).abs().max()) <= 2.1e-4
    for _ in range(59):
        e1.train_step()
        e2.train_step()
    e1.stream.synchronize(); e2.stream.synchronize()
    assert e1.graph_nodes() >= 8
    assert int(e1.step_t.item()) == 60 and int(e2.step_t.item()) == 60
    l1, a1 = e1.read_metrics(0, 60)
    l2, a2 = e2.read_metrics(0, 60)
    assert abs(l1[0] - l2[0]) < 1e-4 and abs(l1[-10:].mean() - l2[-10:].mean()) < 0.05
    assert l1[-10:].mean() < l1[:10].mean() - 0.05
    loss, acc = e1.evaluate(dd1.images, dd1.labels)
    assert np.isfinite(loss) and 0.0 <= acc <= 1.0


@pytest.mark.parametrize("B", [100, 37, 200])
def test_fused_step_matches_unfused_sequence(B):
    """train_step() (one launch sequence: fc GEMMs fused, conv filter gradients fused, W_fc1's Adam update riding in the
    latter, rest of the optimizer last) against fwd_bwd(3) + apply() (gradients materialised, one optimizer launch over
    the whole arena): same parameters and moments up to fp32 atomic ordering, step by step."""
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    dd1, dd2 = _dataset(n=4000, seed=3), _dataset(n=4000, seed=3)
    e1 = NativeMnistEngine(dd1, batch_size=B, seed=5, eval_chunk=500, lr=1e-3, double_softmax=False)
    e2 = NativeMnistEngine(dd2, batch_size=B, seed=5, eval_chunk=500, lr=1e-3, double_softmax=False, use_graph=False)
    assert e1.early_fc1_adam()
    for _ in range(2):
        e1.train_step()
        e2.fwd_bwd(3)
        e2.apply()
        torch.cuda.synchronize()
        dp = float((e1.P - e2.P).abs().max())
        dm = float((e1.M - e2.M).abs().max() / e2.M.abs().max())
        dv = float((e1.V - e2.V).abs().max() / e2.V.abs().max())
        assert dp < 1e-4 and dm < 1e-4 and dv < 1e-4, (dp, dm, dv)  # (one Adam step moves a parameter by up to lr = 1e-3)
    assert int(e1.step_t.item()) == 2 == int(e2.step_t.item())
    # the atomically accumulated gradient segments are re-zeroed by both optimizer paths
    lo = e1.layout["b_fc1"][0]
    assert float(e1.G[lo:].abs().max()) == 0.0 and float(e2.G[lo:].abs().max()) == 0.0


def test_multi_step_graph_matches_single_steps():
    """train_steps() replays several steps (here 8) per graph launch; it must walk through the same batches, dropout masks and Adam
    steps as 8 single launches, also across an epoch boundary (reshuffle between chunks)."""
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    dd1, dd2 = _dataset(n=1500, seed=2), _dataset(n=1500, seed=2)  # 15 steps per epoch
    e1 = NativeMnistEngine(dd1, batch_size=100, seed=11, eval_chunk=500, lr=1e-4)
    e2 = NativeMnistEngine(dd2, batch_size=100, seed=11, eval_chunk=500, lr=1e-4)
    e1.prepare(chunk=8)
    e1.train_steps(37, chunk=8)
    for _ in range(37):
        e2.train_step()
    e1.stream.synchronize(); e2.stream.synchronize()
    assert "full*8" in e1._graphs and int(e1.step_t.item()) == 37 == int(e2.step_t.item()) == e1.global_step
    l1, a1 = e1.read_metrics(0, 37)
    l2, a2 = e2.read_metrics(0, 37)
    assert np.abs(l1 - l2).max() < 5e-3 and np.abs(a1 - a2).max() <= 0.02
    assert torch.equal(dd1.perm, dd2.perm)


def test_eval_and_predict_match_oracle():
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    from distributed_tensorflow_amd.models import mnist_cnn
    dd = _dataset(n=1300, seed=2)
    eng = NativeMnistEngine(dd, batch_size=100, seed=7, eval_chunk=512)
    params = {k: v.clone() for k, v in eng.param_views().items()}
    loss, acc = eng.evaluate(dd.images, dd.labels)
    logits = mnist_cnn.forward_logits(params, dd.images, bf16_sim=True)
    ref_loss, p = mnist_cnn.softmax_xent(logits, dd.labels.long())
    assert abs(loss - float(ref_loss)) < 2e-3
    probs, pred = eng.predict(dd.images[:37])
    assert _relerr(probs, p[:37]) < 2e-2
    agree = float((pred.long() == p[:37].argmax(1)).float().mean())
    assert agree > 0.9
    assert abs(acc - float((p.argmax(1) == dd.labels.long()).float().mean())) < 0.02


def test_training_reaches_high_accuracy_on_synthetic():
    from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
    dd = _dataset(n=6000, seed=4)
    # single softmax trains fast at lr 1e-3; the reference's double softmax needs its lr 1e-4
    # (at 1e-3 the saturated second softmax collapses — the PyTorch oracle shows the same)
    eng = NativeMnistEngine(dd, batch_size=100, seed=9, lr=1e-3, eval_chunk=2000, double_softmax=False)
    for _ in range(300):
        eng.train_step()
    loss, acc = eng.evaluate(dd.images, dd.labels)
    assert acc > 0.95, (loss, acc)
    eng2 = NativeMnistEngine(dd, batch_size=100, seed=9, lr=1e-4, eval_chunk=2000, double_softmax=True)
    for _ in range(600):
        eng2.train_step()
    loss2, acc2 = eng2.evaluate(dd.images, dd.labels)
    assert acc2 > 0.8, (loss2, acc2)


@pytest.mark.parametrize("comm", ["ipc-gather", "ipc-gather-bf16", "ipc-zerocopy", "ipc-inline", "ipc-overlap", "ipc-auto", "rccl"])
def test_native_data_parallel_path_two_ranks_one_gpu(tmp_path, comm):
    """The multi-GPU step with 2 ranks sharing the single test GPU (RCCL refuses duplicate devices, so the process
    group is gloo).  ``ipc-*`` = fwd+bwd | direct IPC all-reduce | Adam as one hipGraph, in each of its schedules
    (gather: fc1 dW computed over all ranks' published operands; zero-copy: fc1 dW produced into the exchange buffer, Adam reads the result buffer; inline: staged copies;
    overlap: fc bucket reduced on a forked stream; auto: picked by the start-up trial); ``rccl`` = the collective path
    (part1 graph -> bucket-A all-reduce || part2 graph -> bucket-B all-reduce -> Adam graph), host-staged for gloo."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dtype = "bf16" if comm.endswith("-bf16") else "fp32"   # (fp32 = the headline engine)
    comm = comm[:-5] if comm.endswith("-bf16") else comm
    port = {"ipc-gather": "29574", "ipc-zerocopy": "29577", "ipc-inline": "29576", "ipc-overlap": "29579", "ipc-auto": "29575",
            "rccl": "29578"}[comm]
    env = dict(os.environ, MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    schedule = None
    if comm.startswith("ipc-"):
        if comm == "ipc-overlap":  # a candidate only where every rank owns a GPU, unless forced
            env["DTF_IPC_FORCE_OVERLAP"] = "1"
        if comm != "ipc-auto":
            schedule = env["DTF_DP_SCHEDULE"] = comm
        comm = "ipc"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5",
           "--backend", "gloo", "--train-size", "4000", "--comm", comm, "--ref-steps", "0", "--dtype", dtype]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 200 and out["replicas_in_sync"] is True
    assert out["config"]["gradient_transport"] == comm
    if schedule is not None:
        assert out["config"]["dp_schedule"] == schedule
    elif comm == "ipc":
        assert out["config"]["dp_schedule"] in out["dp_schedule_us_at_startup"]
    assert out["ipc_wait_timeouts"] in (0, None)
    assert out["final_train_loss"] < 2.35 and out["value"] > 0


_RCCL_ONE_RANK = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch, torch.distributed as dist
from distributed_tensorflow_amd.data import mnist
from distributed_tensorflow_amd.engine.mnist_engine import NativeMnistEngine
from distributed_tensorflow_amd.parallel.ddp import DataParallelTrainer
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
def ds():
    imgs, labels = mnist.synthetic_mnist(3000, seed=1)
    d = mnist.DataSet(mnist._scale_flatten(imgs), mnist.dense_to_one_hot(labels), seed=1)
    return mnist.DeviceDataset(d, "cuda", seed=1)
e1 = NativeMnistEngine(ds(), batch_size=100, seed=5, lr=5e-4)
e2 = NativeMnistEngine(ds(), batch_size=100, seed=5, lr=5e-4)
t1 = DataParallelTrainer(e1, force_collectives=True, overlap=True, comm="rccl")  # part1 | RCCL bucket A || part2 | RCCL bucket B | Adam
t2 = DataParallelTrainer(e2)                                          # plain single-graph step
t1.prepare(); t2.prepare()  # (all graph captures before the first collective is in flight)
for _ in range(40):
    t1.train_step(); t2.train_step()
torch.cuda.synchronize()
l1, _ = e1.read_metrics(0, 40); l2, _ = e2.read_metrics(0, 40)
print(json.dumps({"first": abs(float(l1[0] - l2[0])), "tail": abs(float(l1[-10:].mean() - l2[-10:].mean())),
                  "drop": float(l1[:5].mean() - l1[-5:].mean()), "sync": t1.replicas_in_sync(),
                  "steps": int(e1.step_t.item())}))
dist.destroy_process_group()
"""


def test_rccl_overlapped_step_single_rank(tmp_path):
    """The RCCL (backend "nccl") variant of the overlapped DP step — async all-reduce of arena slices on RCCL's
    stream between the part1/part2/apply graphs — run with one rank, compared against the fused single-graph step."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    r = subprocess.run([sys.executable, str(script), repo], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["steps"] == 40 and out["sync"] is True
    assert out["first"] < 1e-4 and out["tail"] < 0.1 and out["drop"] > 0.0, out
