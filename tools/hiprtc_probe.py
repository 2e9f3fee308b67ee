"""Checks that hiprtc can compile a kernel for the local GPU at run time (needed for the
constraint-program JIT)."""
import ctypes as C, time
rtc = C.CDLL("/opt/rocm/lib/libhiprtc.so")
src = b'extern "C" __global__ void k(unsigned* o, unsigned a){ o[threadIdx.x] = a * threadIdx.x + 1; }'
prog = C.c_void_p()
t0 = time.time()
rc = rtc.hiprtcCreateProgram(C.byref(prog), src, b"k.hip", 0, None, None)
print("create", rc)
opts = (C.c_char_p * 2)(b"--offload-arch=gfx950", b"-O3")
rc = rtc.hiprtcCompileProgram(prog, 2, opts)
print("compile", rc, time.time() - t0)
sz = C.c_size_t()
rtc.hiprtcGetProgramLogSize(prog, C.byref(sz))
log = C.create_string_buffer(sz.value + 1)
rtc.hiprtcGetProgramLog(prog, log)
print("log:", log.value.decode()[:500])
rtc.hiprtcGetCodeSize(prog, C.byref(sz))
print("code size", sz.value)
