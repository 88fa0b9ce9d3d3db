"""roctx ranges around the steps of a sweep (SURVEY.md section 5 "tracing"): a rocprofv3 --marker-trace run shows one range per
(eps, minPts) step.  ctypes over libroctx64; without the library (or with CLOOPS_ROCTX=0) every call is a no-op."""
import ctypes
import os

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("CLOOPS_ROCTX", "1") == "0":
        return None
    for name in ("libroctx64.so", "libroctx64.so.4", os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libroctx64.so")):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.restype = ctypes.c_int
            _lib = lib
            break
        except (OSError, AttributeError):
            continue
    return _lib


class range_(object):
    """with _roctx.range_("sweep step 3: eps 5000 minPts 20"): ..."""
    __slots__ = ("msg", "on")

    def __init__(self, msg):
        self.msg = msg
        self.on = False

    def __enter__(self):
        lib = _load()
        if lib is not None:
            lib.roctxRangePushA(self.msg.encode())
            self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            _lib.roctxRangePop()
        return False
