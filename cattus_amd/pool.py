"""ctypes binding of ``libcattus_pool.so`` (include/cattus_pool.h): record pooling over RCCL for hosts without Python.

The Python path of this repository pools through ``torch.distributed`` (cattus_amd/dist.py); this binding exists for the tests
of the C ABI a Rust / C host would call.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CATTUS_POOL_LIB", _PKG / "libcattus_pool.so"))
ID_BYTES = 128

ABI_SYMBOLS = [
    "cattus_pool_unique_id",
    "cattus_pool_create",
    "cattus_pool_destroy",
    "cattus_pool_records",
    "cattus_pool_reduce_counters",
    "cattus_pool_free",
    "cattus_pool_last_error",
    "cattus_pool_set_timeout",
    "cattus_pool_debug_stalled_collective",
]
E_STATE, E_TIMEOUT = -5, -6  # CATTUS_POOL_E_STATE, CATTUS_POOL_E_TIMEOUT

_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `python -m cattus_amd.build`")
    L = C.CDLL(str(LIB_PATH))
    vp = C.c_void_p
    L.cattus_pool_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.cattus_pool_create.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.cattus_pool_destroy.argtypes = [vp]
    L.cattus_pool_destroy.restype = None
    L.cattus_pool_records.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.cattus_pool_reduce_counters.argtypes = [vp, vp, C.c_uint32]
    L.cattus_pool_free.argtypes = [vp]
    L.cattus_pool_free.restype = None
    L.cattus_pool_last_error.restype = C.c_char_p
    L.cattus_pool_set_timeout.argtypes = [vp, C.c_double]
    L.cattus_pool_debug_stalled_collective.argtypes = [vp]
    _lib = L
    return L


class PoolError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"cattus_pool status {status}: {message}")
        self.status = status


def _check(rc: int):
    if rc != 0:
        raise PoolError(rc, load_library().cattus_pool_last_error().decode(errors="replace"))


def unique_id() -> bytes:
    buf = (C.c_uint8 * ID_BYTES)()
    _check(load_library().cattus_pool_unique_id(buf))
    return bytes(buf)


class Pool:
    """One RCCL communicator: rank `rank` of `world` on HIP device `device`."""

    def __init__(self, uid: bytes, rank: int, world: int, device: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
        _check(self._lib.cattus_pool_create(buf, rank, world, device, C.byref(h)))
        self._h, self.rank = h, rank

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cattus_pool_destroy(self._h)
            self._h = None

    def set_timeout(self, seconds: float):
        """Deadline of every wait inside this pool's entry points: past it they return CATTUS_POOL_E_TIMEOUT (PoolError.status == E_TIMEOUT)
        and the communicator is aborted."""
        _check(self._lib.cattus_pool_set_timeout(self._h, float(seconds)))

    def debug_stalled_collective(self):
        """Diagnostic: a collective that does not complete (what the pooling collectives are to the survivors of a dead peer)."""
        _check(self._lib.cattus_pool_debug_stalled_collective(self._h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def pool_records(self, record_bytes: np.ndarray, record_meta: np.ndarray, record_size: int):
        """-> (bytes [N, R] uint8, meta [N, 3] uint32) sorted by (game, ply) on rank 0, (None, None) elsewhere; and N."""
        rec = np.ascontiguousarray(record_bytes, dtype=np.uint8)
        meta = np.ascontiguousarray(record_meta, dtype=np.uint32)
        ob, om, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(self._lib.cattus_pool_records(self._h, rec.ctypes.data, meta.ctypes.data, len(rec), record_size, C.byref(ob), C.byref(om), C.byref(n)))
        if not ob.value:
            return None, None, n.value
        try:
            b = np.ctypeslib.as_array(C.cast(ob, C.POINTER(C.c_uint8)), shape=(max(1, n.value * record_size),))[: n.value * record_size].copy()
            m = np.ctypeslib.as_array(C.cast(om, C.POINTER(C.c_uint32)), shape=(max(1, n.value * 3),))[: n.value * 3].copy()
        finally:
            self._lib.cattus_pool_free(ob)
            self._lib.cattus_pool_free(om)
        return b.reshape(n.value, record_size), m.reshape(n.value, 3), n.value

    def reduce_counters(self, counters) -> list[int]:
        c = np.ascontiguousarray(counters, dtype=np.uint64)
        _check(self._lib.cattus_pool_reduce_counters(self._h, c.ctypes.data, len(c)))
        return [int(x) for x in c]
