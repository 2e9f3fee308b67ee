"""Probe: can torch (ROCm) wrap a raw device pointer through __cuda_array_interface__?"""
import ctypes, torch

class Ptr:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False),
                                         "version": 2, "strides": None}

a = torch.arange(64, dtype=torch.int32, device="cuda")
p = a.data_ptr()
try:
    t = torch.as_tensor(Ptr(p, 256), device="cuda")
    print("as_tensor ok", t.dtype, t.shape, t.data_ptr() == p)
    t.view(torch.int32)[3] = 777
    torch.cuda.synchronize()
    print("alias", int(a[3]))
except Exception as e:
    print("as_tensor failed:", repr(e))
hip = ctypes.CDLL("libamdhip64.so")
dp = ctypes.c_void_p()
print("hipMalloc", hip.hipMalloc(ctypes.byref(dp), 1024))
t2 = torch.as_tensor(Ptr(dp.value, 1024), device="cuda")
t2.zero_(); t2[5] = 9
torch.cuda.synchronize()
print("foreign ptr ok", int(t2[5]), t2.device)
s = torch.cuda.ExternalStream(torch.cuda.current_stream().cuda_stream)
print("external stream ok", s)
