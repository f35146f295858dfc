"""Device buffers for tests that call the device-pointer entry points of the C ABI, through the HIP runtime the
library itself has loaded (found in /proc/self/maps), so that a test works whether or not torch is in the process.
TEST INFRASTRUCTURE."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        from raytrace_amd import _lib
        _lib.lib()                                   # makes sure a HIP runtime is mapped
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("no HIP runtime is mapped into this process")
        H = C.CDLL(path)
        H.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        H.hipFree.argtypes = [C.c_void_p]
        H.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        H.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        H.hipDeviceSynchronize.argtypes = []
        H.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        H.hipStreamDestroy.argtypes = [C.c_void_p]
        H.hipStreamSynchronize.argtypes = [C.c_void_p]
        H.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
        H.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        H.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        H.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        H.hipGraphExecDestroy.argtypes = [C.c_void_p]
        H.hipGraphDestroy.argtypes = [C.c_void_p]
        _hip = H
    return _hip


class DeviceBuffer:
    def __init__(self, shape, dtype, fill=0):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        if hip().hipMalloc(C.byref(p), max(self.nbytes, 1)) != 0:
            raise RuntimeError("hipMalloc failed")
        self.ptr = p.value
        if hip().hipMemset(self.ptr, fill, max(self.nbytes, 1)) != 0:
            raise RuntimeError("hipMemset failed")
        # the fill runs on the null stream; the library renders on a non-blocking stream of its own, which the null stream
        # does not order: the fill is complete before anything can overwrite it
        if hip().hipDeviceSynchronize() != 0:
            raise RuntimeError("hipDeviceSynchronize failed")

    def fill(self, value):
        if hip().hipMemset(self.ptr, value, max(self.nbytes, 1)) != 0 or hip().hipDeviceSynchronize() != 0:
            raise RuntimeError("hipMemset failed")

    def numpy(self):
        if hip().hipDeviceSynchronize() != 0:
            raise RuntimeError("hipDeviceSynchronize failed")
        out = np.empty(self.shape, self.dtype)
        if self.nbytes and hip().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) != 0:      # hipMemcpyDeviceToHost
            raise RuntimeError("hipMemcpy failed")
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
