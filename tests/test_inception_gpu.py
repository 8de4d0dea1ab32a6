"""Inception-v3 path on the GPU: generic implicit-GEMM conv geometries, pools, preprocessing, the full
batched forward vs the PyTorch oracle, the retrain head kernels and the device sampler."""
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


GEOMS = [  # (H, W, Cin, Cout, kh, kw, stride, padding)
    (35, 35, 192, 64, 1, 1, 1, "SAME"), (35, 35, 48, 64, 5, 5, 1, "SAME"), (35, 35, 288, 384, 3, 3, 2, "VALID"),
    (17, 17, 128, 128, 1, 7, 1, "SAME"), (17, 17, 160, 192, 7, 1, 1, "SAME"), (8, 8, 448, 384, 3, 3, 1, "SAME"),
    (8, 8, 384, 384, 1, 3, 1, "SAME"), (73, 73, 64, 80, 1, 1, 1, "VALID"), (20, 20, 8, 32, 3, 3, 2, "VALID"),
]


@pytest.mark.parametrize("geom", GEOMS)
def test_conv_bias_relu_geometries(geom):
    from distributed_tensorflow_amd.ops import hip_ops
    H, W, Cin, Cout, kh, kw, stride, padding = geom
    torch.manual_seed(hash(geom) % 1000)
    B = 3
    x = torch.randn(B, H, W, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(kh, kw, Cin, Cout, device=DEV) / (kh * kw * Cin) ** 0.5)
    b = torch.randn(Cout, device=DEV) * 0.1
    pad = ((kh - 1) // 2, (kw - 1) // 2) if padding == "SAME" else (0, 0)
    wp = w.permute(3, 0, 1, 2).reshape(Cout, -1).to(torch.bfloat16).contiguous()
    out = hip_ops.conv_bias_relu(x, wp, b, Cin=Cin, KH=kh, KW=kw, stride=stride, pad=pad)
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), wp.float().view(Cout, kh, kw, Cin).permute(0, 3, 1, 2), b,
                          stride=stride, padding=pad)).permute(0, 2, 3, 1)
    assert out.shape == ref.shape
    assert relerr(out, ref) < 1.5e-2


def test_conv_channel_slices_concat_by_write():
    from distributed_tensorflow_amd.ops import hip_ops
    torch.manual_seed(0)
    B, H, W = 2, 17, 17
    x = torch.randn(B, H, W, 96, device=DEV).to(torch.bfloat16)
    w = torch.randn(1, 1, 32, 64, device=DEV) / 6
    wp = w.permute(3, 0, 1, 2).reshape(64, -1).to(torch.bfloat16).contiguous()
    b = torch.zeros(64, device=DEV)
    out = torch.full((B, H, W, 160), 7.0, device=DEV, dtype=torch.bfloat16)
    hip_ops.conv_bias_relu(x, wp, b, Cin=32, KH=1, KW=1, out=out, out_channel_offset=96, in_channel_offset=64)
    ref = F.relu(torch.einsum("bhwc,oc->bhwo", x[..., 64:96].float(), wp.float()))
    assert relerr(out[..., 96:160], ref) < 1.5e-2
    assert float((out[..., :96].float() - 7).abs().max()) == 0.0  # neighbours untouched


def test_pools():
    from distributed_tensorflow_amd import native
    import distributed_tensorflow_amd.data.bottleneck  # noqa: F401 (registers signatures)
    lib = native.hip()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(1)
    B, H, W, C = 2, 35, 35, 64
    x = torch.randn(B, H, W, C, device=DEV).to(torch.bfloat16)
    out = torch.zeros(B, 17, 17, 96, device=DEV, dtype=torch.bfloat16)
    native.check(lib.dtf_maxpool3x3(x.data_ptr(), out.data_ptr() + 2 * 32, B, H, W, C, C, 96, 2, st), "maxpool")
    ref = F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    assert relerr(out[..., 32:96], ref) == 0.0
    out2 = torch.zeros(B, H, W, C, device=DEV, dtype=torch.bfloat16)
    native.check(lib.dtf_avgpool3x3_same(x.data_ptr(), out2.data_ptr(), B, H, W, C, C, C, st), "avgpool")
    ref2 = F.avg_pool2d(x.float().permute(0, 3, 1, 2), 3, 1, 1, count_include_pad=False).permute(0, 2, 3, 1)
    assert relerr(out2, ref2) < 1e-2
    y = torch.randn(B, 8, 8, 2048, device=DEV).to(torch.bfloat16)
    gap = torch.zeros(B, 2048, device=DEV)
    native.check(lib.dtf_global_avgpool(y.data_ptr(), gap.data_ptr(), B, 64, 2048, st), "gap")
    assert relerr(gap, y.float().mean((1, 2))) < 1e-5


def test_preprocess_plain_and_distorted():
    from distributed_tensorflow_amd.data.bottleneck import Distortion, ImageDesc, ImagePreprocessor, preprocess_reference
    rs = np.random.RandomState(0)
    imgs = [rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for h, w in ((480, 640), (100, 100), (333, 211))]
    pre = ImagePreprocessor(DEV)
    out = pre.batch(imgs).float().cpu()
    assert out.shape == (3, 299, 299, 8) and float(out[..., 3:].abs().max()) == 0.0
    for i, a in enumerate(imgs):
        d = ImageDesc(0, a.shape[0], a.shape[1], a.shape[0] / 299.0, a.shape[1] / 299.0, 0.0, 0.0, 0, 1.0)
        ref = preprocess_reference(a, d)
        assert float((out[i] - ref).abs().max()) < 1.6e-2  # bf16 output rounding of values in [-1, 1]
    dist = Distortion(True, 10, 20, 30)
    assert dist.enabled and not Distortion().enabled
    o2 = pre.batch(imgs, dist, random.Random(3)).float()
    assert o2.shape == (3, 299, 299, 8) and torch.isfinite(o2).all()
    assert float((o2.cpu() - out).abs().max()) > 0.05


def test_inception_forward_matches_oracle():
    from distributed_tensorflow_amd.models import inception_v3 as iv3
    net = iv3.InceptionV3(DEV, seed=0)
    assert 21.0e6 < net.num_params < 23.5e6            # ~21.8 M parameters up to pool_3
    assert 11.0e9 < iv3.graph_flops(net.graph) < 12.0e9  # ~11.4 GFLOP / image
    assert len(net.graph.convs) == 94
    torch.manual_seed(0)
    x = (torch.rand(2, 299, 299, 3, device=DEV) * 2 - 1)
    xin = torch.zeros(2, 299, 299, 8, device=DEV, dtype=torch.bfloat16)
    xin[..., :3] = x.to(torch.bfloat16)
    got = net.forward(xin, use_graph=False)
    got_graph = net.forward(xin, use_graph=True)
    ref = iv3.reference_forward(net, xin[..., :3].float().permute(0, 3, 1, 2), bf16_sim=True)
    assert got.shape == (2, 2048) and torch.isfinite(got).all()
    assert float(ref.abs().mean()) > 1e-3   # activations did not die with the random init
    assert relerr(got, ref) < 5e-2
    assert torch.allclose(got, got_graph, rtol=1e-5, atol=1e-6)


def test_graphdef_round_trip_gives_bit_identical_bottlenecks(tmp_path):
    """Export the whole frozen network as a GraphDef, parse it without TensorFlow into a FRESH network: the bottlenecks
    are bit-identical to the exporting network's (``/root/reference/retrain1/retrain.py:470-473``, ``test.py:26-39``)."""
    from distributed_tensorflow_amd.models import inception_v3 as iv3
    from distributed_tensorflow_amd.utils import graphdef as gd
    net = iv3.InceptionV3(DEV, seed=5)
    g = torch.Generator().manual_seed(4)
    for p in net.params.values():
        c = p["beta"].shape[0]
        p["beta"], p["moving_mean"] = torch.randn(c, generator=g) * 0.05, torch.randn(c, generator=g) * 0.05
        p["moving_variance"] = 0.8 + 0.4 * torch.rand(c, generator=g)
    path = str(tmp_path / "g.pb")
    gd.write_inception_graphdef(path, net.graph, net.params, np.zeros((2048, 3), np.float32), np.zeros(3, np.float32))
    layers, _fw, _fb, _name, _ops = gd.read_retrained_graph(path)
    fresh = iv3.InceptionV3(DEV, seed=77)
    fresh.load_tf_weights(layers)
    torch.manual_seed(1)
    xin = torch.zeros(4, 299, 299, 8, device=DEV, dtype=torch.bfloat16)
    xin[..., :3] = (torch.rand(4, 299, 299, 3, device=DEV) * 2 - 1).to(torch.bfloat16)
    a, b = net.forward(xin, use_graph=False), fresh.forward(xin, use_graph=False)
    assert torch.equal(a, b) and float(a.abs().mean()) > 1e-3


def test_bf16_forward_against_the_unrounded_fp32_network():
    """The reference extracts bottlenecks in float32 (``/root/reference/retrain1/retrain.py:228-231``); the native forward
    keeps bf16 operands and activations.  Distance to the UN-rounded fp32 network on the same weights, measured on a
    RANDOM-INIT network with calibrated batch-norm statistics (no pretrained artefact offline; random 94-layer networks
    amplify rounding far more than trained ones): cosine similarity per image 0.986-0.990, relative L2 0.14-0.17 — reported
    and bounded here; parity on the pretrained 2015 weights is unpinned."""
    from distributed_tensorflow_amd.models import inception_v3 as iv3
    from distributed_tensorflow_amd.models.inception_train import calibrate_batchnorm
    net = iv3.InceptionV3(DEV, seed=3)
    torch.manual_seed(7)
    xin = torch.zeros(8, 299, 299, 8, device=DEV, dtype=torch.bfloat16)
    # smooth images (random low-frequency fields), like photographs rather than white noise
    low = torch.rand(8, 3, 19, 19, device=DEV) * 2 - 1
    img = torch.nn.functional.interpolate(low, size=(299, 299), mode="bilinear", align_corners=False)
    xin[..., :3] = img.permute(0, 2, 3, 1).to(torch.bfloat16)
    calibrate_batchnorm(net, xin)   # random-init weights: sane activation scales through all 94 layers
    got = net.forward(xin, use_graph=False)
    ref = iv3.reference_forward(net, xin[..., :3].float().permute(0, 3, 1, 2), bf16_sim=False)   # fp32 weights and activations
    cos = torch.nn.functional.cosine_similarity(got.double(), ref.double(), dim=1)
    rel = ((got.double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1))
    print("bf16 forward vs un-rounded fp32: cosine min %.6f, relative L2 max %.4f" % (float(cos.min()), float(rel.max())))
    assert float(cos.min()) >= 0.975 and float(rel.max()) < 0.25
    # what matters downstream: the features still tell the images apart the way the fp32 network's do (the Gram matrices of
    # the centred features agree)
    gc, rc_ = got.double() - got.double().mean(0), ref.double() - ref.double().mean(0)
    gram_cos = torch.nn.functional.cosine_similarity((gc @ gc.t()).flatten(), (rc_ @ rc_.t()).flatten(), dim=0)
    print("Gram-matrix agreement of the centred features: %.4f" % float(gram_cos))
    assert float(gram_cos) > 0.9


@pytest.mark.parametrize("double_softmax", [True, False])
def test_retrain_head_native_matches_eager(double_softmax):
    from distributed_tensorflow_amd.models.retrain_head import RetrainHead
    torch.manual_seed(2)
    N, C, B = 300, 5, 100
    cache = torch.rand(N, 2048, device=DEV)
    labels = torch.randint(0, C, (N,), device=DEV, dtype=torch.int32)
    rows = torch.randint(0, N, (B,), device=DEV, dtype=torch.int32)
    h_gpu = RetrainHead(C, DEV, learning_rate=0.05, double_softmax=double_softmax, seed=1)
    h_cpu = RetrainHead(C, "cpu", learning_rate=0.05, double_softmax=double_softmax, seed=1)
    h_gpu.weights.copy_(torch.randn(2048, C, device=DEV) * 0.01)
    h_cpu.P.copy_(h_gpu.P.cpu())
    for _ in range(3):
        h_gpu.train_step(cache, labels, rows)
        h_cpu.train_step(cache.cpu(), labels.cpu(), rows.cpu())
        mg, mc = h_gpu.last_metrics(), h_cpu.last_metrics()
        assert abs(mg[0] - mc[0]) < 1e-4 and abs(mg[1] - mc[1]) < 1e-6
    assert relerr(h_gpu.P.cpu(), h_cpu.P) < 1e-4
    ce, acc, probs, pred = h_gpu.evaluate(cache, labels, rows, want_probs=True)
    ce2, acc2, probs2, pred2 = h_cpu.evaluate(cache.cpu(), labels.cpu(), rows.cpu(), want_probs=True)
    assert abs(ce - ce2) < 1e-4 and abs(acc - acc2) < 1e-6 and relerr(probs.cpu(), probs2) < 1e-4


def test_device_sampler_is_class_balanced():
    from distributed_tensorflow_amd.data import image_lists as il
    from distributed_tensorflow_amd.data.bottleneck import BottleneckCache
    counts = np.array([[50, 5, 5], [10, 2, 3], [200, 20, 20]], np.int64)
    offsets = np.zeros_like(counts)
    n = 0
    paths, cls, spl = [], [], []
    for c in range(3):
        for s in range(3):
            offsets[c, s] = n
            n += counts[c, s]
            paths += ["p%d_%d_%d" % (c, s, i) for i in range(counts[c, s])]
            cls += [c] * counts[c, s]
            spl += [s] * counts[c, s]
    idx = il.ImageIndex(["a", "b", "c"], paths, np.asarray(cls, np.int32), np.asarray(spl, np.int32), offsets, counts)
    cache = BottleneckCache(idx, DEV)
    rows = cache.sample_rows("training", 30000, seed=5).cpu().numpy()
    assert np.all(idx.split_id[rows] == 0)
    frac = np.bincount(idx.class_id[rows], minlength=3) / 30000.0
    assert np.all(np.abs(frac - 1 / 3) < 0.02)           # balanced over classes, not over images
    allrows = cache.sample_rows("testing", -1).cpu().numpy()
    assert len(allrows) == 27 and np.all(idx.split_id[allrows] == 1)
    r2 = cache.sample_rows("training", 100, seed=5).cpu().numpy()
    assert not np.array_equal(r2, rows[:100])               # draw counter advances


def test_retrain_end_to_end_inception_small(tmp_path):
    from distributed_tensorflow_amd.apps import retrain
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        rc = retrain.main(["--training_steps", "40", "--synthetic_classes", "3", "--synthetic_images_per_class", "24",
                           "--train_batch_size", "32", "--validation_batch_size", "16", "--eval_step_interval", "20",
                           "--extract_batch_size", "24", "--learning_rate", "0.05", "--seed", "3"], distributed=False)
        assert rc == 0
        assert os.path.exists("retrained_graph.pb")
        assert open("retrained_labels.txt").read().split("\n")[:3] == ["class 00", "class 01", "class 02"]
        assert os.path.isdir("retrain_logs/train") and os.path.isdir("retrain_logs/validation")
    finally:
        os.chdir(cwd)


def test_retrain1_then_test_py_on_held_out_jpegs(tmp_path):
    """The retrained bundle is self-contained (folded Inception backbone + head): ``retrain1/test.py`` loads nothing but
    ``retrained_graph.pb`` / ``retrained_labels.txt`` and classifies JPEGs the training never saw
    (reference ``retrain1/test.py:10-58``)."""
    import subprocess
    import sys
    import numpy as np
    from distributed_tensorflow_amd.data import image_lists as il
    from distributed_tensorflow_amd.utils import checkpoint as ckpt
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w = str(tmp_path)
    env = dict(os.environ, PYTHONPATH=repo)
    r = subprocess.run([sys.executable, "-u", os.path.join(repo, "retrain1", "retrain.py"), "--synthetic",
                        "--synthetic_classes", "3", "--synthetic_images_per_class", "40", "--training_steps", "200",
                        "--eval_step_interval", "50", "--seed", "3"], cwd=w, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-3000:]
    # retrained_graph.pb is a real GraphDef holding Inception-v3 + the new layer (retrain1/retrain.py:470-473)
    from distributed_tensorflow_amd.utils import graphdef as gd
    assert gd.is_graphdef(os.path.join(w, "retrained_graph.pb"))
    layers, fw, fb, final_name, ops = gd.read_retrained_graph(os.path.join(w, "retrained_graph.pb"))
    assert len(layers) == 94 and fw.shape == (2048, 3) and final_name == "final_result"
    assert sum(1 for _n, o, _i in ops if o == "Conv2D") == 94
    held = il.make_synthetic_image_dir(os.path.join(w, "held_out"), num_classes=3, images_per_class=10, seed=4242)
    r2 = subprocess.run([sys.executable, "-u", os.path.join(repo, "retrain1", "test.py"), "--imgs", held], cwd=w, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=200)
    assert r2.returncode == 0, r2.stdout[-3000:]
    files = [l for l in r2.stdout.splitlines() if l.endswith(".jpg")]
    top1 = [l[3:] for l in r2.stdout.splitlines() if l.startswith("=> ")]
    assert len(files) == len(top1) == 30
    truth = [os.path.basename(os.path.dirname(f)).replace("_", " ") for f in files]
    acc = np.mean([a == b for a, b in zip(top1, truth)])
    assert acc >= 0.9, (acc, r.stdout[-1500:])


def test_retrain2_two_ranks_one_gpu(tmp_path):
    """retrain2 (the distributed retrain): 2 ranks sharing the test GPU shard the bottleneck extraction, all-gather the
    cache, all-reduce the head gradients every step (host-staged on the gloo group) and rank 0 exports graph + labels."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    w = str(tmp_path)
    cmd = [sys.executable, "-u", os.path.join(repo, "retrain2", "retrain2.py"), "--num_gpus", "2", "--backend", "gloo",
           "--synthetic", "--synthetic_classes", "3", "--synthetic_images_per_class", "40", "--training_steps", "120",
           "--bottleneck_dir", w + "/bn", "--summaries_dir", w + "/rl", "--output_graph", w + "/rg.pb",
           "--output_labels", w + "/labels.txt", "--model_dir", w + "/model_dir"]
    r = subprocess.run(cmd, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29641"))
    assert r.returncode == 0, r.stdout[-3000:]
    assert "Step 110:" in r.stdout and "(N=100)" in r.stdout and "Final test accuracy" in r.stdout
    acc = float(r.stdout.split("Final test accuracy = ")[1].split("%")[0])
    assert acc >= 80.0, r.stdout[-1500:]
    assert os.path.exists(w + "/rg.pb") and os.path.exists(w + "/labels.txt")


def test_extraction_pipeline_matches_serial_path_on_gpu(tmp_path):
    """decode processes -> pinned ring -> side-stream H2D + resize -> Inception forward fills the HBM cache with the same
    bottlenecks as the serial decode / copy / forward loop (short last batch included), and reports its stage split."""
    from distributed_tensorflow_amd.data import image_lists as il
    from distributed_tensorflow_amd.data.bottleneck import BottleneckCache, ImagePreprocessor
    from distributed_tensorflow_amd.data.pipeline import DecodePipeline
    from distributed_tensorflow_amd.models.inception_v3 import InceptionV3
    d = il.make_synthetic_image_dir(str(tmp_path / "imgs"), num_classes=2, images_per_class=21, seed=9)
    index = il.build_image_index(il.create_image_lists(d, 10, 10), d)
    net = InceptionV3(DEV, seed=2)
    a = BottleneckCache(index, DEV)
    a.extract(net.forward, ImagePreprocessor(DEV), batch_size=16, progress=None)
    pipe = DecodePipeline(DEV, batch_size=16, workers=3, slots=3, chunk=4)
    try:
        b = BottleneckCache(index, DEV)
        n = b.extract(net.forward, ImagePreprocessor(DEV), batch_size=16, progress=None, pipeline=pipe)
        st = b.extract_stats
    finally:
        pipe.close()
    torch.cuda.synchronize()
    assert n == index.num_images == 42 and st["batches"] == 3
    assert torch.allclose(a.values, b.values, rtol=1e-3, atol=1e-3), float((a.values - b.values).abs().max())
    assert float(b.values.abs().sum()) > 0 and {"h2d_ms_per_batch", "gpu_ms_per_batch", "decode_wait_s"} <= set(st)


def test_reference_precision_extraction(tmp_path):
    """``--backbone_precision fp32``: pixels stay fp32 out of the resize kernel, filters / activations / accumulation are
    fp32 (``Fp32Backbone``): the bottlenecks the reference's float32 graph would give for these weights
    (``/root/reference/retrain1/retrain.py:73-82,228-231``).  Pins (a) the fp32 resize kernel against its PyTorch statement,
    (b) the pipelined fp32 extraction against image-by-image ``reference_forward``, (c) the distance of the default bf16
    extraction to it on the same files."""
    from distributed_tensorflow_amd.data import image_lists as il
    from distributed_tensorflow_amd.data.bottleneck import (BottleneckCache, ImageDesc, ImagePreprocessor, decode_jpeg,
                                                            preprocess_reference)
    from distributed_tensorflow_amd.data.pipeline import DecodePipeline
    from distributed_tensorflow_amd.models import inception_v3 as iv3
    from distributed_tensorflow_amd.models.inception_train import calibrate_batchnorm
    d = il.make_synthetic_image_dir(str(tmp_path / "imgs"), num_classes=2, images_per_class=9, seed=4)
    index = il.build_image_index(il.create_image_lists(d, 10, 10), d)
    pre32 = ImagePreprocessor(DEV, fp32=True)
    imgs = [decode_jpeg(p) for p in index.paths[:4]]
    x32 = pre32.batch(imgs)
    assert x32.dtype == torch.float32 and x32.shape == (4, 299, 299, 8) and float(x32[..., 3:].abs().max()) == 0.0
    for i, a in enumerate(imgs):
        ref = preprocess_reference(a, ImageDesc(0, a.shape[0], a.shape[1], a.shape[0] / 299.0, a.shape[1] / 299.0, 0.0, 0.0, 0, 1.0))
        assert float((x32[i].cpu() - ref).abs().max()) < 2e-5
    net = iv3.InceptionV3(DEV, seed=6)
    calibrate_batchnorm(net, ImagePreprocessor(DEV).load(index.paths[:16]))
    fp32 = iv3.Fp32Backbone(net)
    pipe = DecodePipeline(DEV, batch_size=8, workers=2, slots=3, chunk=4, fp32=True)
    try:
        c32 = BottleneckCache(index, DEV)
        n = c32.extract(fp32, pre32, batch_size=8, progress=None, pipeline=pipe)
    finally:
        pipe.close()
    assert n == index.num_images == 18
    # image by image, no pipeline, no weight cache (the library picks other convolution algorithms at batch 1 than at
    # batch 8: fp32 summation orders differ, measured 1.4e-4 through the 94 layers of this random-init network)
    for row in (0, 7, 17):
        x = pre32.load([index.paths[row]])[..., :3].permute(0, 3, 1, 2).contiguous()
        one = iv3.reference_forward(net, x)[0]
        assert relerr(c32.values[row], one) < 1e-3, row
    cbf = BottleneckCache(index, DEV)
    cbf.extract(net.forward, ImagePreprocessor(DEV), batch_size=8, progress=None)
    rel = (cbf.values.double() - c32.values.double()).norm(dim=1) / c32.values.double().norm(dim=1)
    print("bf16 extraction vs fp32 extraction on the same JPEGs: relative L2 max %.4f" % float(rel.max()))
    # (random-init network: rounding is amplified far more than by trained weights, see the test above; 0.29 measured on
    # these 18 synthetic JPEGs)
    assert 1e-4 < float(rel.max()) < 0.4


def test_retrain_cli_with_the_reference_precision_backbone(tmp_path):
    from distributed_tensorflow_amd.apps import retrain
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        rc = retrain.main(["--training_steps", "20", "--synthetic_classes", "2", "--synthetic_images_per_class", "16",
                           "--train_batch_size", "16", "--validation_batch_size", "8", "--eval_step_interval", "10",
                           "--extract_batch_size", "16", "--learning_rate", "0.05", "--seed", "3",
                           "--backbone_precision", "fp32"], distributed=False)
        assert rc == 0 and os.path.exists("retrained_graph.pb")
    finally:
        os.chdir(cwd)


def test_multi_stream_forward_is_bit_identical_to_one_stream(monkeypatch):
    """``StreamPlan`` only reorders launches between streams: same kernels, same operands, same bits — eagerly and as a
    hipGraph replay, with two and with three streams (three: side-to-side edges relayed through the main stream)."""
    from distributed_tensorflow_amd.models import inception_v3 as iv3
    torch.manual_seed(5)
    xin = torch.zeros(16, 299, 299, 8, device=DEV, dtype=torch.bfloat16)
    xin[..., :3] = (torch.rand(16, 299, 299, 3, device=DEV) * 2 - 1).to(torch.bfloat16)
    outs = {}
    for K in (1, 2, 3):
        monkeypatch.setenv("DTF_INCEPTION_STREAMS", str(K))
        net = iv3.InceptionV3(DEV, seed=4, conv_core="v1")   # (one core everywhere: the choice must not differ between the nets)
        outs[K] = (net.forward(xin, use_graph=False), net.forward(xin, use_graph=True), net.forward(xin, use_graph=True))
        assert (net._plans[16].K if K > 1 else not net._plans) and float(outs[K][0].abs().mean()) > 1e-3
    for K in (2, 3):
        for o in outs[K]:
            assert torch.equal(o, outs[1][0]), K
