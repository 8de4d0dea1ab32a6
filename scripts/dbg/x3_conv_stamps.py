"""Diagnostic: phase timestamps of every workgroup of the bf16x6 conv2 forward (shader clock + 100 MHz real-time clock).
Build (CPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDTF_CONV_STAMPS -I csrc/hip csrc/hip/f32x3_conv2.hip -o scripts/dbg/libx3conv_stamps.so"""
import ctypes as C, os
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libx3conv_stamps.so"))
P, I = C.c_void_p, C.c_int
lib.dtf_conv2_fwd_x3.argtypes = [P, P, P, P, P, I, P]
lib.dtf_dbg_set_x3_stamps.argtypes = [P]
B = 100
x = torch.rand(B, 14, 14, 32, device="cuda"); w = torch.randn(800, 64, device="cuda") * 0.05; b = torch.zeros(64, device="cuda")
out = torch.empty(B, 49, 64, device="cuda"); am = torch.empty(B, 49, 64, dtype=torch.uint8, device="cuda")
nwg = (49 * B + 19) // 20
st = torch.zeros(nwg, 16, 2, dtype=torch.int64, device="cuda")
assert lib.dtf_dbg_set_x3_stamps(st.data_ptr()) == 0
for it in range(20):
    assert lib.dtf_conv2_fwd_x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), am.data_ptr(), B, 0) == 0
torch.cuda.synchronize()
t = st.cpu().numpy().astype(np.int64)
n = 7
clk, rt = t[:, :n, 0], t[:, :n, 1]
d = np.diff(clk, axis=1)
print("phases: entry -> image split+stored -> barrier -> k loop done (wave 0) -> barrier -> partial sums stored+barrier -> end")
print("median", np.median(d, axis=0).astype(int), "max", d.max(axis=0))
tot_clk = clk[:, n - 1] - clk[:, 0]; tot_rt = (rt[:, n - 1] - rt[:, 0]) * 10.0
print("workgroup lifetime: median %d cycles = %.2f us -> shader clock %.3f GHz" % (np.median(tot_clk), np.median(tot_rt) / 1e3, np.median(tot_clk / tot_rt)))
print("kernel span by the real-time clock: %.2f us; entry spread %.2f us" % ((rt[:, n - 1].max() - rt[:, 0].min()) / 100.0, (rt[:, 0].max() - rt[:, 0].min()) / 100.0))
full = st.cpu().numpy().astype(np.int64).reshape(nwg, 32)
end = full[:, 16:24] - clk[:, 0:1]
simd = (full[:, 24:32] >> 4) & 3
print("per-wave k-loop end (cycles since workgroup entry), median over workgroups:", np.median(end, axis=0).astype(int))
for g in (0, 1, 100):
    print("wg", g, "simd of waves", simd[g], "end", end[g])
