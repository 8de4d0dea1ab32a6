"""Probe: can shared-memory pages be page-locked for DMA on this box, and what do the alternatives cost?"""
import ctypes, mmap, time
from multiprocessing import shared_memory
import numpy as np, torch
torch.cuda.init()
rt = torch.cuda.cudart()
n = 16 << 20
shm = shared_memory.SharedMemory(create=True, size=n)
t = torch.frombuffer(shm.buf, dtype=torch.uint8, count=n)
for flags in (0, 1, 2, 3):
    rc = rt.cudaHostRegister(t.data_ptr(), n, flags)
    print("shm register flags", flags, "->", int(rc))
    if int(rc) == 0:
        rt.cudaHostUnregister(t.data_ptr()); break
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
rc = hip.hipHostRegister(ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(n), 0)
print("direct hipHostRegister:", rc, hip.hipGetErrorString(rc))
a = np.empty(n, np.uint8); a[:] = 1
rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), 0)
print("numpy heap register:", rc, hip.hipGetErrorString(rc))
m = mmap.mmap(-1, n)
mt = torch.frombuffer(m, dtype=torch.uint8, count=n); mt.fill_(1)
rc = hip.hipHostRegister(ctypes.c_void_p(mt.data_ptr()), ctypes.c_size_t(n), 0)
print("anonymous shared mmap register:", rc, hip.hipGetErrorString(rc))
import resource
print("RLIMIT_MEMLOCK", resource.getrlimit(resource.RLIMIT_MEMLOCK))
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
def rate(src, label):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dev.copy_(src, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%s -> device: %.2f ms (%.1f GB/s)" % (label, dt * 1e3, n / dt / 1e9))
rate(pin, "pinned"); rate(t, "pageable shm")
t0 = time.perf_counter()
for _ in range(10): pin.copy_(t)
dt = (time.perf_counter() - t0) / 10
print("shm -> pinned memcpy: %.2f ms (%.1f GB/s)" % (dt * 1e3, n / dt / 1e9))
del t, mt
shm.close(); shm.unlink()
