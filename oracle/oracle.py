"""ctypes front-end of ``oracle_net.c`` (see that file's header for what it restates)."""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _DIR / "liboracle.so"
    src = _DIR / "oracle_net.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_DIR), "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(str(build()))
        u64p, f32p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.oracle_planes_to_tensor.argtypes = [u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
        L.oracle_planes_to_tensor.restype = C.c_int
        L.oracle_net_create.argtypes = [C.c_void_p, C.c_size_t]
        L.oracle_net_create.restype = C.c_void_p
        L.oracle_net_destroy.argtypes = [C.c_void_p]
        L.oracle_net_destroy.restype = None
        L.oracle_net_forward.argtypes = [C.c_void_p, u64p, C.c_uint32, C.c_uint32, f32p, f32p, C.c_int]
        L.oracle_net_forward.restype = C.c_int
        L.oracle_net_forward_debug.argtypes = [C.c_void_p, u64p, C.c_uint32, f32p, f32p, f32p, f32p]
        L.oracle_net_forward_debug.restype = C.c_int
        L.oracle_net_folded_conv.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p]
        L.oracle_net_folded_conv.restype = C.c_int
        L.oracle_softmax_legal.argtypes = [f32p, u32p, C.c_uint32, f32p]
        L.oracle_softmax_legal.restype = None
        L.oracle_softmax_legal_det.argtypes = [f32p, C.POINTER(C.c_uint16), C.c_uint32, f32p]
        L.oracle_softmax_legal_det.restype = None
        L.oracle_net_eval_cb.argtypes = [C.c_void_p, u64p, C.c_uint32, f32p, f32p]
        L.oracle_net_eval_cb.restype = C.c_int
        L.oracle_tanhf.argtypes = [C.c_float]
        L.oracle_tanhf.restype = C.c_float
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def planes_to_tensor(planes: np.ndarray, board: int, batch: int) -> np.ndarray:
    """planes: uint64 [n, C, W64] -> float32 [batch, C, S, S] (net/mod.rs:121-156)."""
    planes = np.ascontiguousarray(planes, dtype=np.uint64)
    n, c, w64 = planes.shape
    out = np.empty((batch, c, board, board), dtype=np.float32)
    rc = lib().oracle_planes_to_tensor(_p(planes, C.c_uint64), n, c, w64, board, batch, _p(out, C.c_float))
    if rc != 0:
        raise ValueError("invalid sample len")
    return out


class OracleNet:
    def __init__(self, blob: bytes):
        from cattus_amd.weights import parse_header

        self.desc = parse_header(blob)
        self._buf = C.create_string_buffer(blob, len(blob))
        self._h = lib().oracle_net_create(self._buf, len(blob))
        if not self._h:
            raise ValueError("oracle rejected the weight blob")

    def close(self):
        if self._h:
            lib().oracle_net_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def forward(self, planes: np.ndarray, threads: int = 0):
        """planes: uint64 [n, C, W64] -> (policy [n, M] f32, value [n] f32)."""
        d = self.desc
        planes = np.ascontiguousarray(planes, dtype=np.uint64)
        assert planes.ndim == 3 and planes.shape[1] == d.planes and planes.shape[2] * 64 >= d.hw, planes.shape
        n, w64 = planes.shape[0], planes.shape[2]
        policy = np.empty((n, d.moves), dtype=np.float32)
        value = np.empty((n,), dtype=np.float32)
        rc = lib().oracle_net_forward(
            self._h, _p(planes, C.c_uint64), w64, n, _p(policy, C.c_float), _p(value, C.c_float), threads
        )
        assert rc == 0
        return policy, value

    def callback(self, plane_words: int, threads: int = 1):
        """(function address, context address, keepalive) of oracle_net_eval_cb: the oracle behind the
        host library's network-callback signature (cattus_net_eval_fn), no Python in the loop."""

        class Ctx(C.Structure):
            _fields_ = [("net", C.c_void_p), ("w64", C.c_uint32), ("threads", C.c_int32)]

        ctx = Ctx(self._h, plane_words, threads)
        fn = C.cast(lib().oracle_net_eval_cb, C.c_void_p).value
        return fn, C.addressof(ctx), (ctx, self)

    def forward_debug(self, planes_one: np.ndarray):
        d = self.desc
        planes_one = np.ascontiguousarray(planes_one, dtype=np.uint64).reshape(d.planes, -1)
        policy = np.empty((d.moves,), dtype=np.float32)
        value = np.empty((1,), dtype=np.float32)
        stem = np.empty((d.filters, d.board, d.board), dtype=np.float32)
        tower = np.empty((d.filters, d.board, d.board), dtype=np.float32)
        rc = lib().oracle_net_forward_debug(
            self._h,
            _p(planes_one, C.c_uint64),
            planes_one.shape[1],
            _p(policy, C.c_float),
            _p(value, C.c_float),
            _p(stem, C.c_float),
            _p(tower, C.c_float),
        )
        assert rc == 0
        return policy, float(value[0]), stem, tower

    def folded_conv(self, which: int, cin: int):
        d = self.desc
        w = np.empty((9, d.filters, cin), dtype=np.float32)
        b = np.empty((d.filters,), dtype=np.float32)
        rc = lib().oracle_net_folded_conv(self._h, which, _p(w, C.c_float), _p(b, C.c_float))
        assert rc == 0
        return w, b


def softmax_legal(logits: np.ndarray, idx: np.ndarray) -> np.ndarray:
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    out = np.empty((len(idx),), dtype=np.float32)
    lib().oracle_softmax_legal(_p(logits, C.c_float), _p(idx, C.c_uint32), len(idx), _p(out, C.c_float))
    return out


def softmax_legal_det(logits: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """The device softmax of cattus_hip_eval_legal restated (evaluator's exp, move-order sum)."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.uint16)
    out = np.empty((len(idx),), dtype=np.float32)
    lib().oracle_softmax_legal_det(_p(logits, C.c_float), _p(idx, C.c_uint16), len(idx), _p(out, C.c_float))
    return out


def tanhf(x: float) -> float:
    return float(lib().oracle_tanhf(C.c_float(x)))


def default_threads() -> int:
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask
    (a container's mask can list every core of the host while its quota grants a few)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n
