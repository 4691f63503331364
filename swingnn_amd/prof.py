"""Python view of the library's per-kernel profiler (swg_prof_*, include/swingnn_hip.h)."""
import ctypes as C
from contextlib import contextmanager

from . import _lib as L


def read():
    """{kernel name: dict(ms, flops, bytes, launches)} for kernels launched while profiling was enabled."""
    out = {}
    for kid in range(L.lib.swg_prof_kernels()):
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        L.check(L.lib.swg_prof_read(kid, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)), "swg_prof_read")
        if n.value:
            out[L.lib.swg_prof_name(kid).decode()] = dict(ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value)
    return out


@contextmanager
def profiled(serial=True):
    """Per-kernel timing.  serial=True: the part-batches of a split pass are issued one after the other on ONE stream
    meanwhile (engine.network_pass would otherwise overlap the kernels of two part-batches, and each would be timed
    against the other); the launches themselves are those of the product path.  serial=False: the pass is launched
    exactly as the product path launches it (part-batches on their own streams); a kernel's duration then includes
    the time it shares the chip with the other part-batch's kernels -- what the timed region of bench.py sees."""
    from . import engine
    L.lib.swg_prof_reset()
    L.lib.swg_prof_enable(1)
    engine._PROFILING = bool(serial)
    try:
        yield
    finally:
        engine._PROFILING = False
        L.lib.swg_prof_enable(0)
