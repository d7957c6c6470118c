"""Tiny ctypes view of the HIP runtime for tests that need a raw device buffer
(rt_accumulate / rt_resolve take device pointers)."""
import ctypes as C

import numpy as np

import raytracer_amd  # noqa: F401  (loads librt_amd.so, which pulls in the HIP runtime)

_hip = C.CDLL("libamdhip64.so.7")
_hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_hip.hipFree.argtypes = [C.c_void_p]
_hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
_hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_hip.hipDeviceSynchronize.argtypes = []
_hip.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
_hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]


def _ok(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with hipError {code}")


class DeviceBuffer:
    def __init__(self, n_floats):
        self.n = n_floats
        self.ptr = C.c_void_p()
        _ok(_hip.hipMalloc(C.byref(self.ptr), 4 * n_floats), "hipMalloc")
        self.zero()

    def zero(self):
        _ok(_hip.hipMemset(self.ptr, 0, 4 * self.n), "hipMemset")

    def to_host(self):
        out = np.empty(self.n, np.float32)
        _ok(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        _ok(_hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, 4 * self.n, 2), "hipMemcpy D2H")
        return out

    def from_host(self, a):
        a = np.ascontiguousarray(a, np.float32)
        assert a.size == self.n
        _ok(_hip.hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), 4 * self.n, 1), "hipMemcpy H2D")

    def free(self):
        if self.ptr:
            _hip.hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def address(self):
        return self.ptr.value


def device_count():
    n = C.c_int()
    _ok(_hip.hipGetDeviceCount(C.byref(n)), "hipGetDeviceCount")
    return n.value


def free_bytes():
    """Free device memory of the current device."""
    free, total = C.c_size_t(), C.c_size_t()
    _hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    _ok(_hip.hipMemGetInfo(C.byref(free), C.byref(total)), "hipMemGetInfo")
    return free.value


class CapturedGraph:
    """Work captured from a stream into a HIP graph (hipStreamBeginCapture .. hipStreamEndCapture) and replayed."""

    def __init__(self):
        self.stream = C.c_void_p()
        _ok(_hip.hipStreamCreate(C.byref(self.stream)), "hipStreamCreate")
        self.graph, self.executable = C.c_void_p(), C.c_void_p()

    def begin(self):
        _hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
        _ok(_hip.hipStreamBeginCapture(self.stream, 0), "hipStreamBeginCapture")  # hipStreamCaptureModeGlobal

    def end(self):
        _hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _ok(_hip.hipStreamEndCapture(self.stream, C.byref(self.graph)), "hipStreamEndCapture")
        _hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        _ok(_hip.hipGraphInstantiate(C.byref(self.executable), self.graph, None, None, 0), "hipGraphInstantiate")

    def launch(self):
        _hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        _ok(_hip.hipGraphLaunch(self.executable, self.stream), "hipGraphLaunch")
        _ok(_hip.hipStreamSynchronize(self.stream), "hipStreamSynchronize")

    def free(self):
        _hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
        _hip.hipGraphDestroy.argtypes = [C.c_void_p]
        _hip.hipStreamDestroy.argtypes = [C.c_void_p]
        if self.executable:
            _hip.hipGraphExecDestroy(self.executable)
        if self.graph:
            _hip.hipGraphDestroy(self.graph)
        _hip.hipStreamDestroy(self.stream)
