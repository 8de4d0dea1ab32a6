"""Diagnostic: phase timestamps of every workgroup of the fp32 fc backward launch (fc1 data gradient split-K + fc1 dW + fc2 dW).
Build (CPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDTF_GEMM32_STAMPS -I csrc/hip csrc/hip/f32_fc.hip -o scripts/dbg/libf32fc_stamps.so"""
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libf32fc_stamps.so"))
P, I = C.c_void_p, C.c_int
lib.dtf_fc_bwd_trio_f32.argtypes = [P, P, P, I, I, P, P, I, P, I, I, I, P, P, P, I, I, I, I, P, P, I, P]
lib.dtf_dbg_set_trio_stamps.argtypes = [P]
B, splits = 100, 4
dev = "cuda"
pool2 = torch.randn(B, 3136, device=dev); dh = torch.randn(B, 1024, device=dev)
hdrop = torch.randn(B, 1024, device=dev); dz = torch.randn(B, 16, device=dev)
W = torch.randn(3136, 1024, device=dev) * 0.05
dW0 = torch.empty(3136, 1024, device=dev); dW1 = torch.empty(1024, 10, device=dev); dX = torch.zeros(B, 3136, device=dev)
nq, n0, n1 = 1 * 49 * splits, 25 * 16, 8
nwg = nq + n0 + n1
st = torch.zeros(nwg, 64, dtype=torch.int64, device=dev)
lib.dtf_dbg_set_trio_stamps(st.data_ptr())
for it in range(5):
    junk = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()
    dX.zero_(); st.zero_(); torch.cuda.synchronize()
    rc = lib.dtf_fc_bwd_trio_f32(pool2.data_ptr(), dh.data_ptr(), dW0.data_ptr(), 3136, 1024, hdrop.data_ptr(), dz.data_ptr(), 16,
                                 dW1.data_ptr(), 1024, 10, B, dh.data_ptr(), W.data_ptr(), dX.data_ptr(), B, 3136, 1024, splits, None, None, 0, 0)
    torch.cuda.synchronize()
    assert rc == 0, rc
t = st.cpu().numpy().astype(np.int64)
# (the shader clock has a different base on every XCD: only differences within a workgroup mean anything)
for name, lo, hi in (("fc1 data gradient (split-K, atomics)", 0, nq), ("fc1 dW", nq, nq + n0), ("fc2 dW", nq + n0, nwg)):
    tt = t[lo:hi]
    n = int((tt[0] != 0).sum())
    d = np.diff(tt[:, :n], axis=1)
    print("== %s: %d workgroups, %d stamps (entry, [pre-store, post-store] x slabs, loop end, end)" % (name, hi - lo, n))
    print("   median phase lengths (cycles):", np.median(d, axis=0).astype(int))
    life = tt[:, n - 1] - tt[:, 0]
    print("   lifetime median %d max %d cycles; of which before the first slab is staged %d, epilogue %d" % (
        np.median(life), life.max(), np.median(d[:, 0] + d[:, 1]), np.median(d[:, -1])))
print("max err dX", float((dX - dh @ W.t()).abs().max()), "dW0", float((dW0 - pool2.t() @ dh).abs().max()))
