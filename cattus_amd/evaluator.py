"""ctypes binding of ``libcattus_hip.so`` (include/cattus_hip.h).

``HipEvaluator`` plays the role of the reference's ``NNetwork`` + ``Model`` pair
(engine/src/net/mod.rs:14-72, engine/src/net/model.rs:57-218): ``run_net`` takes the leaves of
one batch and returns ``[(logits, value), ...]``; ``planes_to_tensor`` is the stand-alone
tensor packer (engine/src/net/mod.rs:121-156).  There is no CPU implementation behind these
calls: if the shared library is missing or no HIP device is usable they raise.
"""

from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from .weights import NetDesc, parse_header

import os

_PKG = Path(__file__).resolve().parent
# CATTUS_HIP_LIB selects another build of the same ABI (e.g. the stamped diagnostic build)
LIB_PATH = Path(os.environ.get("CATTUS_HIP_LIB", _PKG / "libcattus_hip.so"))

DTYPE_F32, DTYPE_BF16, DTYPE_F16X2, DTYPE_F16 = 0, 1, 2, 3
# "f16x2": the split-precision tower (pairs of f16 values, 22 significant bits; include/cattus_hip.h); "f16": single-term f16
_DTYPES = {"f32": DTYPE_F32, "bf16": DTYPE_BF16, "f16x2": DTYPE_F16X2, "f16": DTYPE_F16}

# every symbol include/cattus_hip.h declares
ABI_SYMBOLS = [
    "cattus_hip_create",
    "cattus_hip_destroy",
    "cattus_hip_desc",
    "cattus_hip_eval",
    "cattus_hip_eval_device",
    "cattus_hip_eval_device_lane",
    "cattus_hip_eval_legal",
    "cattus_hip_lane_stream",
    "cattus_hip_submit",
    "cattus_hip_wait",
    "cattus_hip_flush",
    "cattus_hip_apply",
    "cattus_hip_host_alloc",
    "cattus_hip_host_free",
    "cattus_hip_stats",
    "cattus_hip_time_tower",
    "cattus_hip_mfma_sustained",
    "cattus_hip_planes_to_tensor",
    "cattus_hip_planes_to_tensor_device",
    "cattus_hip_last_error",
    "cattus_hip_version",
    "cattus_hip_runtime_note",
    "cattus_hip_tower_kernel",
    "cattus_hip_create_diag",  # include/cattus_hip_diag.h
]


class CattusHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"cattus_hip status {status}: {message}")
        self.status = status


class EvalConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("max_batch", C.c_uint32),
        ("plane_words", C.c_uint32),
        ("dtype", C.c_uint32),
        ("flush_us", C.c_uint32),
        ("tower_form", C.c_uint32),
    ]


TOWER_FORMS = {"auto": 0, "direct": 1, "winograd": 2}  # cattus_tower_form
# Diagnostic switches (include/cattus_hip_diag.h).  The library reads none of them from the environment; this binding -- the
# harness of the tests and the timing scripts -- forwards the ones it finds there through cattus_hip_create_diag, so that
# `CATTUS_TOWER64=0 python scripts/...` and monkeypatch.setenv keep working.  CATTUS_WINOGRAD=0/1 (rounds 3-4) maps to tower_form.
DIAG_SWITCHES = ("CATTUS_CONV_CB", "CATTUS_CONV_PBW", "CATTUS_FUSED_STEM", "CATTUS_T64_CH", "CATTUS_T64_LS", "CATTUS_SPLIT_W", "CATTUS_T64S_HEADS",
                 "CATTUS_T64S_SHAPE", "CATTUS_TOWER64", "CATTUS_FORCE_GENERIC", "CATTUS_WINO_INPLACE", "CATTUS_ARENA", "CATTUS_WINO_KERNEL",
                 "CATTUS_WINO_PERSIST", "CATTUS_WINO_SPIN")


class Stats(C.Structure):
    _fields_ = [
        ("batches", C.c_uint64),
        ("positions", C.c_uint64),
        ("full_batches", C.c_uint64),
        ("run_seconds_ema", C.c_double),
        ("run_seconds_total", C.c_double),
        ("saturated", C.c_uint64),
    ]


class NetDescC(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("planes", "board", "moves", "blocks", "filters", "vhc", "phc", "fc_hidden")]


_lib = None


def load_library():
    """Load libcattus_hip.so from the package directory; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -m cattus_amd.build` "
            "(there is no CPU fallback for the leaf evaluator)"
        )
    L = C.CDLL(str(LIB_PATH))
    u64p, f32p = C.POINTER(C.c_uint64), C.POINTER(C.c_float)
    vp = C.c_void_p
    L.cattus_hip_create.argtypes = [vp, C.c_size_t, C.POINTER(EvalConfig), C.POINTER(vp)]
    L.cattus_hip_create_diag.argtypes = [vp, C.c_size_t, C.POINTER(EvalConfig), C.c_char_p, C.POINTER(vp)]
    L.cattus_hip_destroy.argtypes = [vp]
    L.cattus_hip_destroy.restype = None
    L.cattus_hip_desc.argtypes = [vp, C.POINTER(NetDescC)]
    L.cattus_hip_eval.argtypes = [vp, u64p, C.c_uint32, f32p, f32p]
    L.cattus_hip_eval_device.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
    L.cattus_hip_eval_device_lane.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp, vp, vp]
    L.cattus_hip_lane_stream.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    u16p = C.POINTER(C.c_uint16)
    L.cattus_hip_eval_legal.argtypes = [vp, u64p, C.c_uint32, u16p, u16p, C.c_uint32, f32p, f32p]
    L.cattus_hip_submit.argtypes = [vp, u64p, C.POINTER(C.c_uint64)]
    L.cattus_hip_wait.argtypes = [vp, C.c_uint64, f32p, f32p]
    L.cattus_hip_flush.argtypes = [vp]
    L.cattus_hip_apply.argtypes = [vp, u64p, C.c_uint32, f32p, f32p]
    L.cattus_hip_host_alloc.argtypes = [C.c_size_t]
    L.cattus_hip_host_alloc.restype = vp
    L.cattus_hip_host_free.argtypes = [vp]
    L.cattus_hip_host_free.restype = None
    L.cattus_hip_stats.argtypes = [vp, C.POINTER(Stats)]
    L.cattus_hip_time_tower.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.cattus_hip_mfma_sustained.argtypes = [vp, C.c_double, C.POINTER(C.c_double)]
    L.cattus_hip_planes_to_tensor.argtypes = [C.c_int, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
    L.cattus_hip_planes_to_tensor_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.cattus_hip_last_error.restype = C.c_char_p
    L.cattus_hip_version.restype = C.c_char_p
    L.cattus_hip_runtime_note.restype = C.c_char_p
    L.cattus_hip_tower_kernel.argtypes = [vp]
    L.cattus_hip_tower_kernel.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("cattus_hip_last_error", "cattus_hip_version", "cattus_hip_runtime_note", "cattus_hip_host_alloc",
                                                   "cattus_hip_tower_kernel"):
            fn.restype = C.c_int
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise CattusHipError(rc, load_library().cattus_hip_last_error().decode(errors="replace"))


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def planes_to_tensor(planes: np.ndarray, board: int, batch_size: int, device: int = 0) -> np.ndarray:
    """HIP drop-in for ``planes_to_tensor`` (engine/src/net/mod.rs:121-156).

    planes: uint64 ``[n, C, plane_words]`` -> float32 ``[batch_size, C, S, S]``, rows ``n..`` zero.
    """
    planes = np.ascontiguousarray(planes, dtype=np.uint64)
    n, c, w64 = planes.shape
    out = np.empty((batch_size, c, board, board), dtype=np.float32)
    _check(load_library().cattus_hip_planes_to_tensor(device, _u64(planes), n, c, w64, board, batch_size, _f32(out)))
    return out


class HipEvaluator:
    """One network resident on one MI355X.

    Mirrors ``NNetwork::new(model_path, inference_cfg, batch_size, cache)`` (net/mod.rs:24-39) with
    the weight blob in place of the model path and ``{"engine": "hip", "device", "dtype"}`` in place
    of ``InferenceConfig``.
    """

    def __init__(
        self,
        blob: bytes,
        batch_size: int,
        plane_words: int,
        dtype: str = "bf16",
        device: int = 0,
        flush_us: int = 200,
        tower_form: str | None = None,
        switches: dict | None = None,
    ):
        """tower_form: "auto" | "direct" | "winograd" (cattus_tower_form; None = "auto", or what CATTUS_WINOGRAD=0/1 in the
        environment says).  switches: diagnostic switches for cattus_hip_create_diag (None = the CATTUS_* ones found in the
        environment, {} = none)."""
        self.desc: NetDesc = parse_header(blob)
        self.batch_size = int(batch_size)
        self.plane_words = int(plane_words)
        self.dtype = dtype
        self.device = device
        self._lib = load_library()
        if tower_form is None:
            legacy = os.environ.get("CATTUS_WINOGRAD")
            tower_form = "auto" if not legacy else ("winograd" if legacy[0] == "1" else "direct")
            if tower_form == "winograd" and not (dtype == "f16x2" and self.desc.board == 8 and self.desc.blocks > 0 and self.desc.filters >= 128 and self.desc.filters % 64 == 0):
                tower_form = "auto"  # the environment form was a wish ("where the shape allows it"); the config field is a demand
        if switches is None:
            switches = {k: os.environ[k] for k in DIAG_SWITCHES if k in os.environ}
        self.tower_form = tower_form
        cfg = EvalConfig(C.sizeof(EvalConfig), device, self.batch_size, self.plane_words, _DTYPES[dtype], flush_us, TOWER_FORMS[tower_form])
        h = C.c_void_p()
        buf = C.create_string_buffer(blob, len(blob))
        if switches:
            text = ";".join(f"{k}={v}" for k, v in switches.items()).encode()
            _check(self._lib.cattus_hip_create_diag(buf, len(blob), C.byref(cfg), text, C.byref(h)))
        else:
            _check(self._lib.cattus_hip_create(buf, len(blob), C.byref(cfg), C.byref(h)))
        self._h = h

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.cattus_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- evaluation -------------------------------------------------------------------------
    def _planes(self, planes) -> np.ndarray:
        planes = np.ascontiguousarray(planes, dtype=np.uint64)
        if planes.ndim != 3 or planes.shape[1:] != (self.desc.planes, self.plane_words):
            raise ValueError(f"planes shape {planes.shape} != [n, {self.desc.planes}, {self.plane_words}]")
        return planes

    def eval(self, planes) -> tuple[np.ndarray, np.ndarray]:
        """planes uint64 ``[n, C, plane_words]`` -> (logits ``[n, M]``, values ``[n]``)."""
        planes = self._planes(planes)
        n = planes.shape[0]
        policy = np.empty((n, self.desc.moves), dtype=np.float32)
        value = np.empty((n,), dtype=np.float32)
        _check(self._lib.cattus_hip_eval(self._h, _u64(planes), n, _f32(policy), _f32(value)))
        return policy, value

    def eval_legal(self, planes, legal_idx, legal_count) -> tuple[np.ndarray, np.ndarray]:
        """``eval`` + ``calc_moves_probs`` (net/mod.rs:100-119) on the device.

        legal_idx uint16 ``[n, L]``, legal_count uint16 ``[n]`` -> (probs ``[n, L]``, values ``[n]``).
        """
        planes = self._planes(planes)
        n = planes.shape[0]
        legal_idx = np.ascontiguousarray(legal_idx, dtype=np.uint16)
        legal_count = np.ascontiguousarray(legal_count, dtype=np.uint16)
        if legal_idx.ndim != 2 or legal_idx.shape[0] != n or legal_count.shape != (n,):
            raise ValueError("legal_idx must be [n, L] and legal_count [n]")
        probs = np.empty(legal_idx.shape, dtype=np.float32)
        value = np.empty((n,), dtype=np.float32)
        u16 = C.POINTER(C.c_uint16)
        _check(
            self._lib.cattus_hip_eval_legal(
                self._h, _u64(planes), n, legal_idx.ctypes.data_as(u16), legal_count.ctypes.data_as(u16),
                legal_idx.shape[1], _f32(probs), _f32(value),
            )
        )
        return probs, value

    def run_net(self, planes) -> list[tuple[np.ndarray, float]]:
        """``NNetwork::run_net`` (net/mod.rs:41-72): one ``(logits, value)`` pair per leaf."""
        policy, value = self.eval(planes)
        return [(policy[i], float(value[i])) for i in range(len(value))]

    def eval_device(self, d_planes: int, n: int, d_policy: int, d_value: int, stream: int = 0, lane: int = 0):
        """Asynchronous evaluation on raw device pointers (all buffers resident in HBM), enqueued on
        ``stream`` (a hipStream_t handle; 0 is HIP's legacy default stream)."""
        _check(self._lib.cattus_hip_eval_device_lane(self._h, lane, d_planes, n, d_policy, d_value, stream))

    def lane_stream(self, lane: int = 0) -> int:
        """Handle of the lane's own stream."""
        st = C.c_void_p()
        _check(self._lib.cattus_hip_lane_stream(self._h, lane, C.byref(st)))
        return st.value or 0

    # -- leaf server (Batcher::apply replacement, util/batch.rs:49-177) ---------------------
    def submit(self, planes_one) -> int:
        p = np.ascontiguousarray(planes_one, dtype=np.uint64).reshape(self.desc.planes, self.plane_words)
        t = C.c_uint64()
        _check(self._lib.cattus_hip_submit(self._h, _u64(p), C.byref(t)))
        return t.value

    def wait(self, ticket: int) -> tuple[np.ndarray, float]:
        policy = np.empty((self.desc.moves,), dtype=np.float32)
        value = C.c_float()
        _check(self._lib.cattus_hip_wait(self._h, ticket, _f32(policy), C.byref(value)))
        return policy, value.value

    def flush(self):
        _check(self._lib.cattus_hip_flush(self._h))

    # -- introspection ----------------------------------------------------------------------
    def stats(self) -> dict:
        s = Stats()
        _check(self._lib.cattus_hip_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in Stats._fields_}

    def tower_kernel(self) -> str:
        """Name of the kernel that runs this evaluator's tower."""
        return self._lib.cattus_hip_tower_kernel(self._h).decode()

    def mfma_sustained(self, seconds: float = 1.0) -> float:
        """TFLOP/s the device's matrix pipe sustains on back-to-back MFMAs of this evaluator's tower kind (diagnostic)."""
        t = C.c_double()
        _check(self._lib.cattus_hip_mfma_sustained(self._h, float(seconds), C.byref(t)))
        return t.value

    def time_tower(self, n: int, reps: int) -> tuple[float, int]:
        """(average device microseconds of one tower conv launch, launches per forward)."""
        us, launches = C.c_float(), C.c_uint32()
        _check(self._lib.cattus_hip_time_tower(self._h, n, reps, C.byref(us), C.byref(launches)))
        return us.value, launches.value
