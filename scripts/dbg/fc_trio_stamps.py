"""Diagnostic: phase timestamps of every workgroup of the fp32 fc backward launch (fc1 data gradient split-K + fc1 dW + fc2 dW).
Build (CPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDTF_GEMM32_STAMPS -I csrc/hip csrc/hip/f32_fc.hip -o scripts/dbg/libf32fc_stamps.so"""
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libf32fc_stamps.so"))
P, I = C.c_void_p, C.c_int
lib.dtf_fc_bwd_trio_f32.argtypes = [P, P, P, I, I, P, P, I, P, I, I, I, P, P, P, I, I, I, I, P, P, I, P]
lib.dtf_dbg_set_trio_stamps.argtypes = [P]
B, splits = 100, int(os.environ.get('SPLITS', '6'))
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
    tt = t[lo:hi, :62]
    n = int((tt[0] != 0).sum())
    d = np.diff(tt[:, :n], axis=1)
    print("== %s: %d workgroups, %d stamps (entry, first slab staged, [next slab stored, slab multiplied + barrier] x slabs, loop end, end)" % (name, hi - lo, n))
    print("   median phase lengths (cycles):", np.median(d, axis=0).astype(int))
    life = tt[:, n - 1] - tt[:, 0]
    print("   lifetime median %d max %d cycles; of which before the first slab is staged %d, epilogue %d" % (
        np.median(life), life.max(), np.median(d[:, 0] + d[:, 1]), np.median(d[:, -1])))
print("max err dX", float((dX - dh @ W.t()).abs().max()), "dW0", float((dW0 - pool2.t() @ dh).abs().max()))

# placement: which CU got which workgroups (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0])
hw, xcc = t[:, 62], t[:, 63] & 15
cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)
ids = {}
for w in range(nwg):
    ids.setdefault(int(cu[w]), []).append(w)
print("distinct CUs used: %d; workgroups per CU: min %d max %d" % (len(ids), min(map(len, ids.values())), max(map(len, ids.values()))))
print("first 24 workgroups -> (xcc, se, sh, cu):", [(int(xcc[w]), int((hw[w] >> 13) & 7), int((hw[w] >> 12) & 1), int((hw[w] >> 8) & 15)) for w in range(24)])
role = lambda w: "q" if w < nq else ("d" if w < nq + n0 else "s")
mix = {}
for c, ws in ids.items():
    k = "".join(sorted(role(w) for w in ws))
    mix[k] = mix.get(k, 0) + 1
print("role mix per CU (q = data gradient, d = fc1 dW, s = fc2 dW):", dict(sorted(mix.items())))
t0 = t[:, 0].astype(np.float64)
print("same-CU workgroup ids (first 6 CUs):", [ids[c] for c in sorted(ids)[:6]])
