"""ctypes binding of ``libcattus_selfplay.so`` (include/cattus_selfplay.h) and the self-play CLI.

``run_self_play`` is the counterpart of the reference's ``<game>_self_player`` binary
(training/self-play/src/self_play_cmd.rs:55-153): same engine JSON, same output files
(``{game:08}_{ply:03}.traindata`` in two directories, a summary JSON), with leaves evaluated on the
MI355X through ``libcattus_hip.so``.  The network is passed to the C++ driver as a raw function
pointer, so no Python runs on the evaluation path.

    python -m cattus_amd.selfplay --game chess --model1-path m.cattus --model2-path m.cattus \
        --games-num 64 --out-dir1 d1 --out-dir2 d2 --summary-file s.json --config-file engine.json
"""

from __future__ import annotations

import ctypes as C
import json
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
# CATTUS_SELFPLAY_LIB selects another build of the same ABI (e.g. one with -DCATTUS_SCHED_STATS)
LIB_PATH = Path(os.environ.get("CATTUS_SELFPLAY_LIB", _PKG / "libcattus_selfplay.so"))

GAMES = {"tictactoe": 0, "ttt": 0, "hex4": 1, "hex5": 2, "hex7": 3, "hex9": 4, "hex11": 5, "chess": 6}

NET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float))
NET_LEGAL_FN = C.CFUNCTYPE(
    C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_uint32,
    C.POINTER(C.c_float), C.POINTER(C.c_float),
)

ABI_SYMBOLS = [
    "cattus_sp_game_info",
    "cattus_sp_run",
    "cattus_sp_result_summary",
    "cattus_sp_result_records",
    "cattus_sp_result_free",
    "cattus_sp_last_error",
    "cattus_sp_stub_net",
    "cattus_sp_trace_game",
    "cattus_sp_trace_game_ex",
    "cattus_sp_play_moves",
    "cattus_sp_test_dirichlet",
    "cattus_sp_test_temperature_choice",
    "cattus_sp_pos_new",
    "cattus_sp_pos_free",
    "cattus_sp_pos_status",
    "cattus_sp_pos_turn",
    "cattus_sp_pos_legal",
    "cattus_sp_pos_moved",
    "cattus_sp_pos_flipped",
    "cattus_sp_pos_equal",
    "cattus_sp_pos_planes",
    "cattus_sp_pos_str",
    "cattus_sp_pos_flipped_move_nn",
    "cattus_sp_pos_test_record",
    "cattus_sp_chess_perft",
    "cattus_sp_chess_nn_moves",
]


class SpConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("sim_num", C.c_uint32),
        ("explore_factor", C.c_float),
        ("temperature_count", C.c_uint32),
        ("temperature_threshold", C.c_uint32 * 8),
        ("temperature_value", C.c_float * 8),
        ("prior_noise_alpha", C.c_float),
        ("prior_noise_epsilon", C.c_float),
        ("cache_size", C.c_uint64),
        ("batch_size", C.c_uint32),
        ("threads", C.c_uint32),
        ("concurrent_games", C.c_uint32),
        ("seed", C.c_uint64),
        ("first_game", C.c_uint32),
        ("game_stride", C.c_uint32),
        ("host_alloc", C.c_void_p),
        ("host_free", C.c_void_p),
        ("legal_net1", C.c_void_p),
        ("legal_net2", C.c_void_p),
        ("eval_threads", C.c_uint32),
        ("leaves_in_flight", C.c_uint32),
        ("max_game_plies", C.c_uint32),
        ("game_list", C.POINTER(C.c_uint32)),
        ("progress_path", C.c_char_p),
    ]


class SpSummary(C.Structure):
    _fields_ = [
        ("player1_wins", C.c_uint32),
        ("player2_wins", C.c_uint32),
        ("draws", C.c_uint32),
        ("positions", C.c_uint64),
        ("records", C.c_uint64),
        ("activation_count", C.c_uint64),
        ("node_evals", C.c_uint64),
        ("cache_hits", C.c_uint64),
        ("cache_misses", C.c_uint64),
        ("run_duration", C.c_double),
        ("search_duration", C.c_double),
        ("seconds", C.c_double),
        ("steady_seconds", C.c_double),
        ("steady_node_evals", C.c_uint64),
        ("adjudicated", C.c_uint64),
        ("steady_plies", C.c_uint64),
        ("steady_batches", C.c_uint64),
    ]


_lib = None


def available_cpus() -> int:
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `python -m cattus_amd.build`")
    # worker threads sleep at the end of a round instead of spinning: the evaluator's server thread and the
    # second slot group need the cores while a batch is on the GPU
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")
    L = C.CDLL(str(LIB_PATH))
    vp = C.c_void_p
    L.cattus_sp_game_info.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    L.cattus_sp_run.argtypes = [C.c_int, C.POINTER(SpConfig), vp, vp, vp, vp, C.c_uint32, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.cattus_sp_result_summary.argtypes = [vp, C.POINTER(SpSummary)]
    L.cattus_sp_result_records.argtypes = [vp, vp, vp]
    L.cattus_sp_result_free.argtypes = [vp]
    L.cattus_sp_result_free.restype = None
    L.cattus_sp_last_error.restype = C.c_char_p
    L.cattus_sp_trace_game.argtypes = [C.c_int, C.POINTER(SpConfig), vp, vp, C.c_uint32, vp, C.c_size_t]
    L.cattus_sp_trace_game.restype = C.c_int64
    L.cattus_sp_trace_game_ex.argtypes = [C.c_int, C.POINTER(SpConfig), vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, C.c_size_t]
    L.cattus_sp_trace_game_ex.restype = C.c_int64
    L.cattus_sp_play_moves.argtypes = [C.c_int, vp, C.c_uint32, C.POINTER(C.c_uint32)]
    L.cattus_sp_test_dirichlet.argtypes = [C.c_uint64, C.c_float, C.c_uint32, C.c_uint32, vp]
    L.cattus_sp_test_temperature_choice.argtypes = [C.c_uint64, vp, C.c_uint32, C.c_float, C.c_uint32, vp]
    L.cattus_sp_pos_new.argtypes = [C.c_int, C.c_char_p]
    L.cattus_sp_pos_new.restype = vp
    L.cattus_sp_pos_free.argtypes = [vp]
    L.cattus_sp_pos_free.restype = None
    for n in ("status", "turn"):
        getattr(L, f"cattus_sp_pos_{n}").argtypes = [vp]
    L.cattus_sp_pos_legal.argtypes = [vp, vp, vp, C.c_int]
    L.cattus_sp_pos_moved.argtypes = [vp, C.c_int]
    L.cattus_sp_pos_moved.restype = vp
    L.cattus_sp_pos_flipped.argtypes = [vp]
    L.cattus_sp_pos_flipped.restype = vp
    L.cattus_sp_pos_equal.argtypes = [vp, vp]
    L.cattus_sp_pos_planes.argtypes = [vp, vp]
    L.cattus_sp_pos_str.argtypes = [vp, C.c_char_p, C.c_int]
    L.cattus_sp_pos_flipped_move_nn.argtypes = [vp, C.c_int]
    L.cattus_sp_pos_test_record.argtypes = [vp, vp, C.c_int]
    L.cattus_sp_chess_perft.argtypes = [C.c_char_p, C.c_int]
    L.cattus_sp_chess_perft.restype = C.c_uint64
    L.cattus_sp_chess_nn_moves.argtypes = [vp]
    _lib = L
    return L


def game_info(game: str) -> dict:
    out = (C.c_uint32 * 5)()
    if load_library().cattus_sp_game_info(GAMES[game], out) != 0:
        raise ValueError(game)
    return dict(board=out[0], moves=out[1], planes=out[2], plane_words=out[3], record_bytes=out[4])


def make_config(
    sim_num: int,
    explore_factor: float = 1.41421,
    temperature_policy=((9999, 0.0),),
    prior_noise_alpha: float = 0.0,
    prior_noise_epsilon: float = 0.0,
    cache_size: int = 1000,
    batch_size: int = 1,
    threads: int = 1,
    concurrent_games: int = 0,
    seed: int = 1,
    first_game: int = 0,
    game_stride: int = 1,
    eval_threads: int = 0,
    leaves_in_flight: int = 1,
    max_game_plies: int = 0,
    game_list=None,
    progress_path=None,
) -> SpConfig:
    c = SpConfig()
    c.struct_size = C.sizeof(SpConfig)
    c.sim_num, c.explore_factor = sim_num, explore_factor
    tp = list(temperature_policy)
    assert 1 <= len(tp) <= 8, "temperature_policy must hold 1..8 entries"
    c.temperature_count = len(tp)
    for i, (thr, val) in enumerate(tp):
        c.temperature_threshold[i], c.temperature_value[i] = int(thr), float(val)
    c.prior_noise_alpha, c.prior_noise_epsilon = prior_noise_alpha, prior_noise_epsilon
    c.cache_size, c.batch_size, c.threads = cache_size, batch_size, max(1, min(threads, available_cpus()))
    c.concurrent_games, c.seed, c.first_game, c.game_stride = concurrent_games, seed, first_game, game_stride
    c.eval_threads = eval_threads  # 0 = the driver's default (2 batches in flight)
    c.max_game_plies = max_game_plies  # > 0: adjudicate a draw after that many plies (bounded samples; not in the reference)
    c.leaves_in_flight = leaves_in_flight  # > 1: several leaves per tree at the network (virtual loss), not in the reference
    # failure containment (include/cattus_selfplay.h): explicit global game indices (a re-queue) and the progress file
    if game_list is not None:
        c._game_list = np.ascontiguousarray(list(game_list), dtype=np.uint32)  # kept alive by the struct object
        c.game_list = c._game_list.ctypes.data_as(C.POINTER(C.c_uint32))
    if progress_path is not None:
        c.progress_path = os.fsencode(str(progress_path))
    return c


def config_from_engine_json(cfg: dict, **overrides) -> SpConfig:
    """The reference's engine JSON (self_play_cmd.rs:34-53): {model:{inference,batch_size}, mcts:{...}, threads}."""
    m = cfg["mcts"]
    kw = dict(
        sim_num=m["sim_num"],
        explore_factor=m["explore_factor"],
        temperature_policy=[tuple(x) for x in m["temperature_policy"]],
        prior_noise_alpha=m["prior_noise_alpha"],
        prior_noise_epsilon=m["prior_noise_epsilon"],
        cache_size=m["cache_size"],
        batch_size=cfg["model"]["batch_size"],
        threads=cfg["threads"],
        leaves_in_flight=m.get("leaves_in_flight", 1),  # extension: not a key of the reference's JSON
    )
    kw.update(overrides)
    return make_config(**kw)


class Net:
    """A (function pointer, context) pair for the C++ driver."""

    def __init__(self, fn_addr: int, ctx: int, keepalive=None):
        self.fn_addr, self.ctx, self._keep = fn_addr, ctx, keepalive
        self.legal_addr = None  # a cattus_net_eval_legal_fn: the network also does calc_moves_probs

    @staticmethod
    def stub(game: str) -> "Net":
        L = load_library()
        info = game_info(game)
        ctx = (C.c_uint32 * 2)(info["moves"], info["planes"] * info["plane_words"])
        return Net(C.cast(L.cattus_sp_stub_net, C.c_void_p).value, C.addressof(ctx), keepalive=ctx)

    @staticmethod
    def hip(evaluator, device_softmax: bool = False) -> "Net":
        """libcattus_hip's cattus_hip_eval has exactly the callback signature; ctx = evaluator handle.

        device_softmax: use cattus_hip_eval_legal, i.e. the softmax over each leaf's legal moves
        (net/mod.rs:100-119) runs on the GPU and only those probabilities cross PCIe.
        """
        fn = C.cast(evaluator._lib.cattus_hip_eval, C.c_void_p).value
        net = Net(fn, evaluator._h.value, keepalive=evaluator)
        if device_softmax:
            net.legal_addr = C.cast(evaluator._lib.cattus_hip_eval_legal, C.c_void_p).value
        # page-locked batch buffers: the evaluator then transfers without a staging copy
        net.host_alloc = C.cast(evaluator._lib.cattus_hip_host_alloc, C.c_void_p).value
        net.host_free = C.cast(evaluator._lib.cattus_hip_host_free, C.c_void_p).value
        return net

    @staticmethod
    def hip_batched(evaluator) -> "Net":
        """cattus_hip_apply: a blocking call per search thread, batched across threads by the evaluator's
        leaf server -- the reference's threading model (one thread per game meeting in Batcher::apply)."""
        fn = C.cast(evaluator._lib.cattus_hip_apply, C.c_void_p).value
        return Net(fn, evaluator._h.value, keepalive=evaluator)

    @staticmethod
    def raw(fn_addr: int, ctx: int, keepalive=None) -> "Net":
        """Any C function with the cattus_net_eval_fn signature (e.g. the CPU oracle's callback, tests / bench baseline)."""
        return Net(fn_addr, ctx, keepalive=keepalive)

    @staticmethod
    def python(fn) -> "Net":
        """Wrap ``fn(planes uint64 [n, words]) -> (policy [n, M], value [n])`` (tests only)."""

        def trampoline(_ctx, planes_p, n, policy_p, value_p):
            try:
                words = Net._words[id(cb)]
                planes = np.ctypeslib.as_array(planes_p, shape=(n, words)).copy()
                policy, value = fn(planes)
                policy = np.ascontiguousarray(policy, dtype=np.float32)
                value = np.ascontiguousarray(value, dtype=np.float32)
                C.memmove(policy_p, policy.ctypes.data, policy.nbytes)
                C.memmove(value_p, value.ctypes.data, value.nbytes)
                return 0
            except Exception:  # pragma: no cover - surfaced as a status code
                import traceback

                traceback.print_exc()
                return -99

        cb = NET_FN(trampoline)
        net = Net(C.cast(cb, C.c_void_p).value, 0, keepalive=cb)
        net._cb = cb
        return net

    @staticmethod
    def python_legal(fn) -> "Net":
        """Wrap ``fn(planes [n, words], legal_idx [n, L], legal_count [n]) -> (probs [n, L], value [n])`` (tests only)."""

        def trampoline(_ctx, planes_p, n, idx_p, cnt_p, stride, probs_p, value_p):
            try:
                words = Net._words[id(cb)]
                planes = np.ctypeslib.as_array(planes_p, shape=(n, words)).copy()
                idx = np.ctypeslib.as_array(idx_p, shape=(n, stride)).copy()
                cnt = np.ctypeslib.as_array(cnt_p, shape=(n,)).copy()
                probs, value = fn(planes, idx, cnt)
                probs = np.ascontiguousarray(probs, dtype=np.float32)
                value = np.ascontiguousarray(value, dtype=np.float32)
                assert probs.shape == (n, stride)
                C.memmove(probs_p, probs.ctypes.data, probs.nbytes)
                C.memmove(value_p, value.ctypes.data, value.nbytes)
                return 0
            except Exception:  # pragma: no cover - surfaced as a status code
                import traceback

                traceback.print_exc()
                return -99

        cb = NET_LEGAL_FN(trampoline)
        stub = NET_FN(lambda *a: -98)  # never called: the driver uses legal_addr
        net = Net(C.cast(stub, C.c_void_p).value, 0, keepalive=(cb, stub))
        net.legal_addr = C.cast(cb, C.c_void_p).value
        net._cb = cb
        return net

    _words: dict = {}

    def bind_words(self, words: int):
        if hasattr(self, "_cb"):
            Net._words[id(self._cb)] = words
        return self


def _err() -> str:
    return load_library().cattus_sp_last_error().decode(errors="replace")


def run_self_play(game: str, cfg: SpConfig, net1: Net, net2: Net | None, games_num: int, out_dir1=None, out_dir2=None,
                  keep_records: bool = True) -> dict:
    L = load_library()
    info = game_info(game)
    for n in (net1, net2):
        if n is not None:
            n.bind_words(info["planes"] * info["plane_words"])
    for d in (out_dir1, out_dir2):
        if d is not None:
            os.makedirs(d, exist_ok=True)
    if getattr(net1, "host_alloc", None) and (net2 is None or getattr(net2, "host_alloc", None)):
        cfg.host_alloc, cfg.host_free = net1.host_alloc, net1.host_free
    cfg.legal_net1 = net1.legal_addr
    cfg.legal_net2 = net2.legal_addr if net2 is not None else None
    if getattr(cfg, "_game_list", None) is not None and len(cfg._game_list) != games_num:
        raise ValueError(f"game_list holds {len(cfg._game_list)} games, games_num is {games_num}")
    res = C.c_void_p()
    rc = L.cattus_sp_run(
        GAMES[game], C.byref(cfg), net1.fn_addr, net1.ctx, net2.fn_addr if net2 else None, net2.ctx if net2 else None,
        games_num, str(out_dir1).encode() if out_dir1 else None, str(out_dir2).encode() if out_dir2 else None,
        1 if keep_records else 0, C.byref(res),
    )
    if rc != 0:
        raise RuntimeError(f"self-play failed ({rc}): {_err()}")
    try:
        s = SpSummary()
        L.cattus_sp_result_summary(res, C.byref(s))
        out = {name: getattr(s, name) for name, _ in SpSummary._fields_}
        n = int(s.records) if keep_records else 0
        rec = np.zeros((n, info["record_bytes"]), dtype=np.uint8)
        meta = np.zeros((n, 3), dtype=np.uint32)
        if n:
            L.cattus_sp_result_records(res, rec.ctypes.data, meta.ctypes.data)
        out["record_bytes"], out["record_meta"] = rec, meta
        return out
    finally:
        L.cattus_sp_result_free(res)


def trace_game(game: str, cfg: SpConfig, net: Net, max_plies: int = 512, forced=(), search_from: int = 0):
    """One self-play game; returns [(chosen nn_idx, [(nn_idx, visits), ...]), ...] per searched ply.

    forced: policy indices played at plies 0.. instead of the search's choice (the choice is still
    reported); plies below search_from are played without a search."""
    L = load_library()
    info = game_info(game)
    net.bind_words(info["planes"] * info["plane_words"])
    cap = 2 + max_plies * (2 + 2 * 256)
    buf = np.zeros(cap, dtype=np.uint32)
    fm = np.ascontiguousarray(list(forced), dtype=np.uint16)
    n = L.cattus_sp_trace_game_ex(GAMES[game], C.byref(cfg), net.fn_addr, net.ctx, max_plies, fm.ctypes.data if len(fm) else None,
                                  len(fm), search_from, buf.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("trace_game failed: " + _err())
    plies, w, out = int(buf[0]), 1, []
    for _ in range(plies):
        chosen, k = int(buf[w]), int(buf[w + 1])
        pairs = [(int(buf[w + 2 + 2 * i]), int(buf[w + 3 + 2 * i])) for i in range(k)]
        out.append((chosen, pairs))
        w += 2 + 2 * k
    return out


def play_moves(game: str, moves) -> tuple:
    """Game::play_single_turn over policy-index moves from the initial position -> (status, plies played);
    status 'ongoing' or the winner +1 / -1 / 0 (0 also for a draw by repetition)."""
    mv = np.ascontiguousarray(list(moves), dtype=np.uint16)
    played = C.c_uint32()
    st = load_library().cattus_sp_play_moves(GAMES[game], mv.ctypes.data if len(mv) else None, len(mv), C.byref(played))
    if st <= -100:
        raise ValueError(_err())
    return ("ongoing" if st == 2 else st), played.value


class Position:
    """Rule-level handle used by the tests (engine/src/game/mod.rs Position trait)."""

    def __init__(self, game: str, s: str | None = None, _h=None):
        self.game = game
        self._L = load_library()
        self._h = _h if _h is not None else self._L.cattus_sp_pos_new(GAMES[game], s.encode() if s is not None else None)
        if not self._h:
            raise ValueError(f"bad position: {s!r}")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.cattus_sp_pos_free(self._h)
            self._h = None

    def status(self):
        """'ongoing' or the winner as +1 / -1 / 0."""
        s = self._L.cattus_sp_pos_status(self._h)
        return "ongoing" if s == 2 else s

    def turn(self) -> int:
        return self._L.cattus_sp_pos_turn(self._h)

    def legal_moves(self):
        nn = (C.c_uint16 * 256)()
        names = C.create_string_buffer(256 * 12)
        n = self._L.cattus_sp_pos_legal(self._h, nn, names, 256)
        return [(names.raw[12 * i : 12 * i + 12].split(b"\0")[0].decode(), int(nn[i])) for i in range(n)]

    def moved(self, k: int) -> "Position":
        h = self._L.cattus_sp_pos_moved(self._h, k)
        if not h:
            raise IndexError(k)
        return Position(self.game, _h=h)

    def play(self, name: str) -> "Position":
        for k, (nm, _) in enumerate(self.legal_moves()):
            if nm == name:
                return self.moved(k)
        raise ValueError(f"illegal move {name}")

    def flipped(self) -> "Position":
        return Position(self.game, _h=self._L.cattus_sp_pos_flipped(self._h))

    def flipped_move_nn(self, k: int) -> int:
        return self._L.cattus_sp_pos_flipped_move_nn(self._h, k)

    def __eq__(self, other):
        return bool(self._L.cattus_sp_pos_equal(self._h, other._h))

    def planes(self) -> np.ndarray:
        info = game_info(self.game)
        out = np.zeros((info["planes"], info["plane_words"]), dtype=np.uint64)
        self._L.cattus_sp_pos_planes(self._h, out.ctypes.data)
        return out

    def __str__(self):
        buf = C.create_string_buffer(256)
        self._L.cattus_sp_pos_str(self._h, buf, 256)
        return buf.value.decode()

    def test_record(self) -> bytes:
        info = game_info(self.game)
        buf = (C.c_uint8 * info["record_bytes"])()
        n = self._L.cattus_sp_pos_test_record(self._h, buf, info["record_bytes"])
        if n < 0:
            raise ValueError("position must have Player1 to move")
        return bytes(buf)


def dirichlet_draws(seed: int, alpha: float, k: int, draws: int) -> np.ndarray:
    """`draws` samples [draws, k] of the search's root-noise distribution Dir(alpha, ..., alpha) (mcts/mod.rs:419-446)."""
    out = np.zeros((draws, k), dtype=np.float64)
    if load_library().cattus_sp_test_dirichlet(seed, alpha, k, draws, out.ctypes.data) != 0:
        raise ValueError("bad arguments")
    return out


def temperature_choice_counts(seed: int, probs, temperature: float, draws: int) -> np.ndarray:
    """How often each of the moves with visit probabilities `probs` is chosen at `temperature` > 0 (mcts/mod.rs:403-415)."""
    p = np.ascontiguousarray(probs, dtype=np.float32)
    counts = np.zeros(len(p), dtype=np.uint32)
    if load_library().cattus_sp_test_temperature_choice(seed, p.ctypes.data, len(p), temperature, draws, counts.ctypes.data) != 0:
        raise ValueError("bad arguments")
    return counts


def chess_perft(fen: str, depth: int) -> int:
    return int(load_library().cattus_sp_chess_perft(fen.encode(), depth))


def chess_nn_moves() -> list[str]:
    buf = C.create_string_buffer(1880 * 8)
    load_library().cattus_sp_chess_nn_moves(buf)
    return [buf.raw[8 * i : 8 * i + 8].split(b"\0")[0].decode() for i in range(1880)]


def write_summary(path, summary: dict):
    """Summary JSON of the reference (self_play_cmd.rs:111-150), same metric names."""
    out = {
        "player1_wins": summary["player1_wins"],
        "player2_wins": summary["player2_wins"],
        "draws": summary["draws"],
        "metrics": {
            "model.activation_count": summary["activation_count"],
            "model.run_duration": summary["run_duration"],
            "mcts.search_duration": summary["search_duration"],
            "cache.hits": summary["cache_hits"],
            "cache.misses": summary["cache_misses"],
            "node_evals": summary["node_evals"],
            "evals_per_sec": summary["node_evals"] / max(summary["seconds"], 1e-9),
            "games_seconds": summary["seconds"],
            # f16 towers: activation values clamped at the f16 range (0 for every network inside it; include/cattus_hip.h)
            "model.saturated": summary.get("saturated", 0),
        },
    }
    with open(path, "x") as f:  # create_new, as the reference
        json.dump(out, f)


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="MI355X self-play (counterpart of the reference's <game>_self_player)")
    ap.add_argument("--game", required=True, choices=sorted(GAMES))
    ap.add_argument("--model1-path", required=True)
    ap.add_argument("--model2-path", required=True)
    ap.add_argument("--games-num", type=int, required=True)
    ap.add_argument("--out-dir1", required=True)
    ap.add_argument("--out-dir2", required=True)
    ap.add_argument("--summary-file")
    ap.add_argument("--config-file", required=True)
    args = ap.parse_args(argv)

    from .evaluator import HipEvaluator

    with open(args.config_file) as f:
        engine = json.load(f)
    inf = engine["model"].get("inference", {})
    if inf.get("engine", "hip") != "hip":
        raise SystemExit(f"this self-player only implements the 'hip' inference engine, got {inf.get('engine')!r}")
    info = game_info(args.game)
    cfg = config_from_engine_json(engine, concurrent_games=engine.get("concurrent_games", 0))

    def load(path):
        blob = Path(path).read_bytes()
        return HipEvaluator(blob, batch_size=engine["model"]["batch_size"], plane_words=info["plane_words"],
                            dtype=inf.get("dtype", "f16x2"), device=inf.get("device", 0))

    ev1 = load(args.model1_path)
    same = os.path.abspath(args.model1_path) == os.path.abspath(args.model2_path)
    ev2 = None if same else load(args.model2_path)
    res = run_self_play(args.game, cfg, Net.hip(ev1), None if same else Net.hip(ev2), args.games_num, args.out_dir1,
                        args.out_dir2, keep_records=False)
    # the f16 towers clamp what leaves the f16 range (cattus_stats.saturated): say so once, and put it in the summary
    res["saturated"] = sum(int(ev.stats().get("saturated", 0)) for ev in (ev1, ev2) if ev is not None)
    if res["saturated"]:
        import sys

        print(f"{Path(sys.argv[0]).name}: WARNING: {res['saturated']} activation values left the f16 range and were clamped: the "
              f"outputs of dtype {inf.get('dtype', 'f16x2')!r} are not this network's -- use \"dtype\": \"f32\" in model.inference", file=sys.stderr)
    if args.summary_file:
        write_summary(args.summary_file, res)


if __name__ == "__main__":
    main()
