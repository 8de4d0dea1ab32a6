"""Diagnostic: stand-alone times of the roles of the fp32 fc backward launch (split-K data gradient, fc1 dW, the fused trio)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from distributed_tensorflow_amd import native
lib = native.hip()
P, I = C.c_void_p, C.c_int
lib.dtf_fc_bwd_trio_f32.argtypes = [P, P, P, I, I, P, P, I, P, I, I, I, P, P, P, I, I, I, I, P, P, I, P]
lib.dtf_fc_bwd_data_f32.argtypes = [P, P, P, I, I, I, I, P]
lib.dtf_fc_dw_f32.argtypes = [P, I, P, I, P, I, I, I, I, P]
B = 100
dev = "cuda"
pool2 = torch.randn(B, 3136, device=dev); dh = torch.randn(B, 1024, device=dev)
hdrop = torch.randn(B, 1024, device=dev); dz = torch.randn(B, 16, device=dev)
W = torch.randn(3136, 1024, device=dev) * 0.05
dW0 = torch.empty(3136, 1024, device=dev); dW1 = torch.empty(1024, 10, device=dev); dX = torch.zeros(B, 3136, device=dev)
junk = torch.empty(128 << 20, dtype=torch.float32, device=dev)


def timeit(name, fn, n=200, flush=False):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(n):
        if flush:
            junk.add_(1.0)  # evicts L2 / MALL
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        t += a.elapsed_time(b)
    print("%-46s %7.2f us%s" % (name, 1000 * t / n, "  (cold caches)" if flush else ""))


for flush in (False, True):
    for s in (2, 4, 8):
        timeit("data gradient alone, %d splits" % s, lambda: lib.dtf_fc_bwd_data_f32(dh.data_ptr(), W.data_ptr(), dX.data_ptr(), B, 3136, 1024, s, None), flush=flush)
    timeit("fc1 dW alone", lambda: lib.dtf_fc_dw_f32(pool2.data_ptr(), 3136, dh.data_ptr(), 1024, dW0.data_ptr(), 3136, 1024, B, 1, None), flush=flush)
    timeit("trio, 4 splits", lambda: lib.dtf_fc_bwd_trio_f32(pool2.data_ptr(), dh.data_ptr(), dW0.data_ptr(), 3136, 1024, hdrop.data_ptr(), dz.data_ptr(), 16, dW1.data_ptr(), 1024, 10, B, dh.data_ptr(), W.data_ptr(), dX.data_ptr(), B, 3136, 1024, 4, None, None, 0, None), flush=flush)
    timeit("empty (event overhead)", lambda: None, flush=flush)
