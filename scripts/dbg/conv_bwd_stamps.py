"""Diagnostic: phase timestamps of every workgroup of the fp32 conv2 data-gradient and filter-gradient kernels.
Build (CPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDTF_CONV_STAMPS -I csrc/hip csrc/hip/f32_conv2.hip -o scripts/dbg/libf32conv_stamps.so"""
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libf32conv_stamps.so"))
P, I = C.c_void_p, C.c_int
lib.dtf_conv2_bwd_data_f32.argtypes = [P, P, P, P, P, P, I, P, P]
lib.dtf_conv_dw_duo_f32.argtypes = [P, P, P, P, P, P, I, P, P, P, I, P, P, P, P, P, P]
lib.dtf_dbg_set_conv_stamps.argtypes = [P]
B = 100
dev = "cuda"
dpool2 = torch.randn(B, 49, 64, device=dev); pool2 = torch.rand(B, 49, 64, device=dev) - 0.3
amax2 = torch.randint(0, 4, (B, 49, 64), dtype=torch.uint8, device=dev)
W = torch.randn(25, 32, 64, device=dev) * 0.05
pool1 = torch.rand(B, 196, 32, device=dev) - 0.3; dpool1 = torch.empty(B, 196, 32, device=dev)
db2 = torch.zeros(64, device=dev); dW2 = torch.zeros(25, 32, 64, device=dev)


def report(name, nwg, n, labels, run, extra=()):
    st = torch.zeros(2 * nwg + 8, 32, dtype=torch.int64, device=dev)
    assert lib.dtf_dbg_set_conv_stamps(st.data_ptr()) == 0
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    t = st.cpu().numpy().astype(np.int64)[:nwg].reshape(nwg, 16, 2)
    clk, rt = t[:, :n, 0], t[:, :n, 1]
    d = np.diff(clk, axis=1)
    print("==", name, "(%d workgroups)" % nwg)
    print("phases:", labels)
    print("  median", np.median(d, axis=0).astype(int), " max", d.max(axis=0), "(shader cycles)")
    full = st.cpu().numpy().astype(np.int64)[:nwg].reshape(nwg, 16, 2)
    if extra:
        allst = st.cpu().numpy().astype(np.int64)
        pw = allst[nwg:2 * nwg]
        end = pw[:, :8] - clk[:, 0:1]
        simd = (pw[:, 16:24] >> 4) & 3
        print("  per-wave k-loop end (cycles since entry), median over workgroups:", np.median(end, axis=0).astype(int))
        for g in (0, 1, 100):
            print("   wg", g, "simd of waves", simd[g], "end", end[g])
    for i, lab in extra:
        print("  %s: median %d cycles after entry" % (lab, np.median(full[:, i, 0] - full[:, 0, 0])))
    tot_clk = clk[:, n - 1] - clk[:, 0]; tot_rt = (rt[:, n - 1] - rt[:, 0]) * 10.0
    print("  workgroup lifetime: median %d cycles = %.2f us (clock %.3f GHz)" % (np.median(tot_clk), np.median(tot_rt) / 1e3, np.median(tot_clk / np.maximum(tot_rt, 1))))
    print("  kernel span by the real-time clock: %.2f us; entry spread %.2f us; last exit - median exit %.2f us" % (
        (rt[:, n - 1].max() - rt[:, 0].min()) / 100.0, (rt[:, 0].max() - rt[:, 0].min()) / 100.0,
        (rt[:, n - 1].max() - np.median(rt[:, n - 1])) / 100.0))


report("conv2 bwd-data", (196 * B + 79) // 80, 5, "entry -> dY staged -> k loop done -> barrier -> end",
       lambda: lib.dtf_conv2_bwd_data_f32(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), W.data_ptr(), pool1.data_ptr(),
                                          dpool1.data_ptr(), B, None, 0),
       extra=((5, "loads issued"), (6, "zero fill + barrier"), (7, "un-pool stores issued")))
report("conv2 dW (conv2 role only)", ((B + 3) // 4) * 10, 6, "entry -> image 0 staged -> image 1 staged -> all k loops done -> barrier -> end",
       lambda: lib.dtf_conv_dw_duo_f32(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), pool1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), B,
                                       None, None, None, 0, None, None, None, None, None, 0))


# ---- the whole launch as the engine issues it: conv2 dW + conv1 dW + W_fc1's Adam update
from distributed_tensorflow_amd import native
images = torch.rand(4000, 784, device=dev); perm = torch.randperm(4000, device=dev, dtype=torch.int32)
step = torch.zeros(1, dtype=torch.int32, device=dev)
amax1 = torch.randint(0, 4, (B, 196, 32), dtype=torch.uint8, device=dev)
dW1 = torch.zeros(25, 32, device=dev); db1 = torch.zeros(32, device=dev)
n = 3136 * 1024
Pm, Gm, Mm, Vm = (torch.zeros(n, device=dev) for _ in range(4))
op = native.OptimParams()
op.P, op.G, op.M, op.V = Pm.data_ptr(), Gm.data_ptr(), Mm.data_ptr(), Vm.data_ptr()
op.n, op.zero_from, op.lr, op.beta1, op.beta2, op.eps, op.grad_scale = n, n, 1e-4, 0.9, 0.999, 1e-8, 1.0
op.step = step.data_ptr()
n2, n1 = ((B + 3) // 4) * 10, min(B, 128)
nwg = 512
st = torch.zeros(2 * nwg + 8, 32, dtype=torch.int64, device=dev)
assert lib.dtf_dbg_set_conv_stamps(st.data_ptr()) == 0
for _ in range(20):
    rc = lib.dtf_conv_dw_duo_f32(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), pool1.data_ptr(), dW2.data_ptr(), db2.data_ptr(), B,
                                 images.data_ptr(), perm.data_ptr(), step.data_ptr(), 40, dpool1.data_ptr(), amax1.data_ptr(),
                                 dW1.data_ptr(), db1.data_ptr(), C.byref(op), 0)
    assert rc == 0, rc
torch.cuda.synchronize()
t = st.cpu().numpy().astype(np.int64)[:nwg].reshape(nwg, 16, 2)
rt0 = t[:, 0, 1][t[:, 0, 1] > 0].min()
for name, lo, hi in (("conv2 dW", 0, n2), ("conv1 dW", n2, n2 + n1), ("Adam W_fc1", n2 + n1, nwg)):
    e, x = (t[lo:hi, 0, 1] - rt0) / 100.0, (t[lo:hi, 5, 1] - rt0) / 100.0
    print("%-10s %3d workgroups: entry %.2f .. %.2f us (median %.2f), exit %.2f .. %.2f us (median %.2f), lifetime median %.2f us" % (
        name, hi - lo, e.min(), e.max(), np.median(e), x.min(), x.max(), np.median(x), np.median(x - e)))
