"""Diagnostic: phase timestamps (shader clock) of every workgroup of the fp32 fc1 forward GEMM.
Build first (CPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDTF_GEMM32_STAMPS -I csrc/hip csrc/hip/f32_fc.hip -o scripts/dbg/libf32fc_stamps.so"""
import ctypes as C, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libf32fc_stamps.so"))
P, I = C.c_void_p, C.c_int
lib.dtf_fc_fwd_f32_stamps.argtypes = [P, P, P, I, I, I, I, P, P]
M, N, K, S = 100, 1024, 3136, 14
x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
out = torch.empty(S, M, N, device="cuda")
nwg = 1 * 16 * S
st = torch.zeros(nwg, 64, dtype=torch.int64, device="cuda")
for it in range(5):
    # evict W from the caches the way a training step does (Adam touches 52 MB in between)
    junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda").normal_()
    st.zero_()
    torch.cuda.synchronize()
    rc = lib.dtf_fc_fwd_f32_stamps(x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, S, st.data_ptr(), 0)
    torch.cuda.synchronize()
    assert rc == 0
t = st.cpu().numpy().astype(np.int64)
n = int((t[0] != 0).sum())
t0 = t[:, 0].min()
rel = t[:, :n] - t0
print("stamps per workgroup:", n, "(entry, [pre-store, post-store] x slabs, pre-epilogue, end)")
print("median over workgroups of each stamp (shader cycles since the first workgroup's entry):")
print(np.median(rel, axis=0).astype(int))
print("deltas:", np.diff(np.median(rel, axis=0)).astype(int))
print("entry: min %d max %d; end: min %d max %d" % (rel[:, 0].min(), rel[:, 0].max(), rel[:, n - 1].min(), rel[:, n - 1].max()))
err = float((out.sum(0) - x @ w).abs().max())
print("max abs err vs torch", err)
# wave-specialised bf16x6 form: the staging side's end-of-slab clocks at [32 ..] (0 when not present)
if (t[:, 32] != 0).any():
    ns = int((t[0, 32:] != 0).sum())
    srel = t[:, 32:32 + ns] - t[:, 0:1]
    mrel = t[:, :n] - t[:, 0:1]
    print("multiplying wave 0, cycles since its entry (entry, staged, [products issued, barrier passed] x slabs, end):")
    print(np.median(mrel, axis=0).astype(int))
    print("staging wave, end of each slab's split + store (cycles since workgroup entry):")
    print(np.median(srel, axis=0).astype(int))
