"""conv2 backward pieces alone (no hosted optimizer, no conv1 role): exact-fp32 MFMA vs bf16x6, CUDA-event timing."""
import torch
from distributed_tensorflow_amd import native
from distributed_tensorflow_amd.ops import hip_ops as ops

dev = torch.device("cuda")
B = 100
torch.manual_seed(0)
pool1 = torch.relu(torch.randn(B, 14, 14, 32, device=dev))
w = torch.randn(5, 5, 32, 64, device=dev) * 0.05
b = torch.randn(64, device=dev) * 0.1
lib = native.hip()


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


for emu in (0, 1):
    lib.dtf_f32_emulation_set(emu)
    pool2, amax2 = ops.conv2_forward_f32(pool1, w, b)
    dpool2 = torch.randn(B, 7, 7, 64, device=dev)
    dW = torch.zeros(5, 5, 32, 64, device=dev)
    db = torch.zeros(64, device=dev)
    dpool1 = torch.empty(B, 14, 14, 32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def dw():
        native.check(lib.dtf_conv_dw_duo_f32(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), pool1.data_ptr(),
                                             dW.data_ptr(), db.data_ptr(), B, None, None, None, 1, None, None, None, None,
                                             None, st), "dw")

    def bd():
        native.check(lib.dtf_conv2_bwd_data_f32(dpool2.data_ptr(), pool2.data_ptr(), amax2.data_ptr(), w.data_ptr(),
                                                pool1.data_ptr(), dpool1.data_ptr(), B, None, st), "bd")

    def fw():
        ops.conv2_forward_f32(pool1, w, b)

    print("emulation=%d  conv2 dW alone %.1f us   bwd-data alone (+pack when emulated) %.1f us   fwd alone (+pack, +alloc) %.1f us"
          % (emu, timeit(dw), timeit(bd), timeit(fw)))
