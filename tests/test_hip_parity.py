"""GPU parity tests: the HIP leaf evaluator (through the C ABI) against the CPU oracle.

Bars (DESIGN.md "Parity"):
  * planes_to_tensor: bit-exact (values are 0.0 / 1.0).
  * dtype f32: bit-exact against the oracle (same fmaf-chain order on both sides) and within the
    reference's cross-runtime tolerance of the reference network's own outputs (tests/golden).
  * dtype f16x2 (split precision, the product default): within the reference's cross-runtime tolerance of the
    reference network's outputs on every fixture, and as close to the float64 run of the reference network as an
    f32 runtime is (bounds F16X2_* below).
  * dtype bf16: bf16 operands / f32 accumulation; per-fixture bounds in BF16_MEASURED below (2x the measured error).
  * dtype f16: single-term f16 operands; per-fixture bounds in F16_MEASURED (2x the measured error, a tenth of bf16's).
  * dtype f16x2 in Winograd form (evaluators of 8x8-board networks with max_batch > 128): the f16x2 bars.
  * per-leaf results never depend on batch size, slot or neighbours (all dtypes, bit-exact).
"""

import threading

import numpy as np
import pytest

from cattus_amd import synth
from cattus_amd.evaluator import CattusHipError, HipEvaluator, planes_to_tensor
from cattus_amd.weights import CHESS, NetDesc, hex_game, seeded_blob
from oracle import oracle

from helpers import blob_for, golden_names, outputs_equal_ref_tol

pytestmark = pytest.mark.gpu

# bf16 tower against the reference network's outputs: measured max |dlogit|, max |dvalue| per fixture (round 3,
# scripts/split_check.py -> profiles/r03_split_check.json); the tests allow twice that, so a precision regression
# goes red.  On the 256- / 512-leaf synthetic batches of the full-size tests the maxima over that many leaves are
# larger than over a fixture's few: BF16_FULL[net] = 2x the measured 0.00842 / 0.00214 (chess 20x256, 256 leaves) and
# 0.0260 / 0.00582 (chess 40x384, 512 leaves).
BF16_MEASURED = {
    "chess_1x1": (4.6e-4, 1e-6), "chess_20x256": (7.5e-3, 1.15e-3), "chess_2x64": (2.2e-3, 3.5e-4), "chess_7x16": (3.9e-3, 2.0e-4),
    "hex11_1x1": (4.0e-4, 5.7e-5), "hex11_2x8": (1.1e-3, 3.5e-4), "hex4_7x16": (2.5e-3, 2.6e-4), "hex7_6x64": (3.2e-3, 5.8e-4),
    "ttt_1x1": (2.3e-4, 1e-6), "ttt_2x64": (1.9e-3, 7.6e-4), "ttt_5x8": (2.7e-3, 7.4e-4), "chess_4x128": (4.0e-3, 6.0e-4),
}
BF16_FULL = {"20x256": (1.7e-2, 4.3e-3), "40x384": (5.2e-2, 1.2e-2)}
# single-term f16 tower, the same way (round 4, profiles/r04_split_check.json): a tenth of bf16's error
F16_MEASURED = {
    "chess_1x1": (3.2e-5, 1e-6), "chess_20x256": (7.5e-4, 7.6e-5), "chess_2x64": (2.1e-4, 4.0e-5), "chess_7x16": (3.7e-4, 3.0e-5),
    "hex11_1x1": (2.3e-5, 1e-6), "hex11_2x8": (7.2e-5, 1.3e-5), "hex4_7x16": (2.3e-4, 5.0e-5), "hex7_6x64": (3.7e-4, 5.6e-5),
    "ttt_1x1": (1e-6, 1e-6), "ttt_2x64": (1.4e-4, 8.4e-5), "ttt_5x8": (2.3e-4, 5.4e-5), "chess_4x128": (4.0e-4, 6.0e-5),
}
# split precision against the float64 run of the reference network: measured max |dlogit| 7.4e-7, |dvalue| 2.1e-7 (chess
# 20x256; the bit-exact f32 tower: 7.7e-7 / 1.9e-7, the reference's own f32 run: 3.3e-7 / 4.9e-8)
F16X2_POLICY_ATOL_VS_F64, F16X2_VALUE_ATOL_VS_F64 = 1.5e-6, 5e-7

# every fixture runs on the MFMA tower: filters are padded to 64 channels, boards above 8x8 take 128 pixel slots
MFMA_SHAPES = golden_names()


def _plane_words(planes):
    return planes.shape[2]


@pytest.mark.parametrize("name", golden_names())
def test_planes_to_tensor_bit_exact(name):
    d, blob, z = blob_for(name)
    planes = z["planes"]
    n = len(planes)
    for batch in (n, n + 5):
        got = planes_to_tensor(planes, d.board, batch)
        want = oracle.planes_to_tensor(planes, d.board, batch)
        assert got.shape == want.shape
        assert (got == want).all()
    assert (got[:n] == z["input_tensor"].astype(np.float32)).all()


def test_planes_to_tensor_large_and_errors():
    planes = synth.random_chess_planes(300, 7)
    got = planes_to_tensor(planes, 8, 512)
    assert (got == oracle.planes_to_tensor(planes, 8, 512)).all()
    with pytest.raises(CattusHipError):
        planes_to_tensor(planes, 8, 299)  # n > batch: the reference asserts (net/mod.rs:122-127)


@pytest.mark.parametrize("name", golden_names())
def test_f32_bit_exact_vs_oracle_and_reference_tolerance(name):
    d, blob, z = blob_for(name)
    planes = z["planes"]
    net = oracle.OracleNet(blob)
    want_p, want_v = net.forward(planes)
    with HipEvaluator(blob, batch_size=len(planes) + 3, plane_words=_plane_words(planes), dtype="f32") as ev:
        got_p, got_v = ev.eval(planes)
    assert (got_p == want_p).all(), f"max |dp| = {np.abs(got_p - want_p).max()}"
    assert (got_v == want_v).all(), f"max |dv| = {np.abs(got_v - want_v).max()}"
    assert outputs_equal_ref_tol(got_p, got_v, z["policy"], z["value"])  # the 20-block net too


@pytest.mark.parametrize("name", MFMA_SHAPES)
def test_bf16_within_stated_tolerance(name):
    d, blob, z = blob_for(name)
    planes = z["planes"]
    with HipEvaluator(blob, batch_size=len(planes), plane_words=_plane_words(planes), dtype="bf16") as ev:
        got_p, got_v = ev.eval(planes)
    ref_p, ref_v = z["policy"], z["value"]
    assert np.isfinite(got_p).all() and np.isfinite(got_v).all()
    dp, dv = BF16_MEASURED[name]
    assert np.abs(got_p - ref_p).max() <= 2 * dp, np.abs(got_p - ref_p).max()
    assert np.abs(got_v - ref_v).max() <= 2 * dv, np.abs(got_v - ref_v).max()


@pytest.mark.parametrize("name", MFMA_SHAPES)
def test_f16_within_stated_tolerance(name):
    """The single-term f16 tower (dtype f16: f16 operands with per-channel pre-scaled weights, f32 accumulation, f32 heads) against the
    reference network's outputs: twice the measured error per fixture, which is a tenth of the bf16 tower's."""
    d, blob, z = blob_for(name)
    planes = z["planes"]
    with HipEvaluator(blob, batch_size=len(planes), plane_words=_plane_words(planes), dtype="f16") as ev:
        got_p, got_v = ev.eval(planes)
        again_p, again_v = ev.eval(planes[:1])
        assert ev.stats()["saturated"] == 0
    assert (again_p[0] == got_p[0]).all() and again_v[0] == got_v[0]  # a leaf's bits do not depend on the batch
    dp, dv = F16_MEASURED[name]
    assert np.abs(got_p - z["policy"]).max() <= 2 * dp, np.abs(got_p - z["policy"]).max()
    assert np.abs(got_v - z["value"]).max() <= 2 * dv, np.abs(got_v - z["value"]).max()
    bdp, bdv = BF16_MEASURED[name]
    assert dp <= bdp and dv <= bdv  # never worse than bf16


@pytest.mark.parametrize("name", golden_names())
def test_f16x2_within_the_reference_tolerance(name):
    """The split-precision tower (pairs of f16 values, three MFMA terms per product, f32 heads) against the outputs of
    the reference's own network (tests/golden, made with training/cattus_train/net_utils.py): inside the reference's
    cross-runtime bar (training/tests/test_net_output.py:28-33: policy rtol 1e-3 / atol 1e-6, value rel 1e-5 / abs
    1e-6) on EVERY fixture, the 20-block one included, and as close to the reference's float64 run as f32 arithmetic gets."""
    d, blob, z = blob_for(name)
    planes = z["planes"]
    with HipEvaluator(blob, batch_size=len(planes) + 2, plane_words=_plane_words(planes), dtype="f16x2") as ev:
        got_p, got_v = ev.eval(planes)
    assert outputs_equal_ref_tol(got_p, got_v, z["policy"], z["value"])
    assert np.abs(got_p - z["policy_f64"]).max() <= F16X2_POLICY_ATOL_VS_F64, np.abs(got_p - z["policy_f64"]).max()
    assert np.abs(got_v - z["value_f64"]).max() <= F16X2_VALUE_ATOL_VS_F64, np.abs(got_v - z["value_f64"]).max()


@pytest.mark.parametrize("game,desc,words,n", [
    ("hex7", dict(**hex_game(7), blocks=6, filters=64, vhc=16, phc=16), 2, 128),    # BASELINE config 2
    ("hex7", dict(**hex_game(7), blocks=3, filters=64, vhc=16, phc=16), 2, 1100),   # many workgroups, ragged batch
    ("chess", dict(**CHESS, blocks=7, filters=16, vhc=8, phc=8), 1, 300),           # the reference's chess net: padded channels
    ("hex11", dict(**hex_game(11), blocks=2, filters=8, vhc=4, phc=4), 2, 37),      # 128-slot boards
    ("hex9", dict(**hex_game(9), blocks=2, filters=40, vhc=16, phc=16), 2, 600),    # 128-slot boards, 600 of them
    ("hex5", dict(**hex_game(5), blocks=4, filters=96, vhc=8, phc=8), 2, 77),       # 96 filters: padded to 128, two cout slabs
    ("ttt", dict(planes=3, board=3, moves=9, blocks=5, filters=8, vhc=8, phc=8), 1, 5),
])
def test_f16x2_tracks_the_f32_tower_on_every_shape(game, desc, words, n):
    """The split tower on the shapes the other towers are tested on -- 64- and 128-slot boards, padded channels, several
    output-channel slabs, ragged batches, both workgroup tiles: every leaf inside the reference's cross-runtime bar of the
    bit-exact f32 tower's result, and independent of the batch it came in."""
    d = NetDesc(**desc)
    blob = seeded_blob(d, 17)
    rng = np.random.default_rng(5)
    hw = d.board * d.board
    planes = np.zeros((n, d.planes, words), dtype=np.uint64)
    bits = rng.integers(0, 2, size=(n, d.planes, hw), dtype=np.uint64)
    for i in range(hw):
        planes[:, :, i >> 6] |= bits[:, :, i] << np.uint64(i & 63)
    with HipEvaluator(blob, batch_size=n, plane_words=words, dtype="f32") as ev:
        want_p, want_v = ev.eval(planes)
    with HipEvaluator(blob, batch_size=n + 3, plane_words=words, dtype="f16x2") as ev:
        got_p, got_v = ev.eval(planes)
        k = max(1, n // 3)
        sub_p, sub_v = ev.eval(planes[k : 2 * k + 1])
    assert outputs_equal_ref_tol(got_p, got_v, want_p, want_v), (np.abs(got_p - want_p).max(), np.abs(got_v - want_v).max())
    assert (sub_p == got_p[k : 2 * k + 1]).all() and (sub_v == got_v[k : 2 * k + 1]).all()


@pytest.mark.parametrize("game,desc,words,n", [
    ("chess", dict(**CHESS, blocks=3, filters=256, vhc=8, phc=8), 1, 256),          # the headline layer shape, full grid
    ("chess", dict(**CHESS, blocks=2, filters=128, vhc=8, phc=8), 1, 61),           # small grid: 128 rows x 32 couts per workgroup
    ("chess", dict(**CHESS, blocks=2, filters=256, vhc=8, phc=8), 1, 100),          # 256 rows x 32 couts per workgroup
    ("hex7", dict(**hex_game(7), blocks=3, filters=64, vhc=16, phc=16), 2, 1100),   # two chunks per layer: the shortest loop
    ("hex11", dict(**hex_game(11), blocks=2, filters=96, vhc=4, phc=4), 2, 37),     # 128-slot boards, padded filters, 128-row workgroups
    ("hex9", dict(**hex_game(9), blocks=2, filters=128, vhc=4, phc=4), 2, 100),     # 128-slot boards, 256-row workgroups
    ("chess", dict(**CHESS, blocks=0, filters=64, vhc=8, phc=8), 1, 5),             # stem only (one chunk)
])
def test_f16x2_weight_paths_agree_bit_for_bit(game, desc, words, n, monkeypatch):
    """The two split-conv kernels -- weights in a register ring fed from L2 (the default), weights through the LDS ring
    (CATTUS_SPLIT_W=0) -- issue the same MFMA sequence per accumulator: their towers agree bit for bit."""
    d = NetDesc(**desc)
    blob = seeded_blob(d, 23)
    rng = np.random.default_rng(8)
    hw = d.board * d.board
    planes = np.zeros((n, d.planes, words), dtype=np.uint64)
    bits = rng.integers(0, 2, size=(n, d.planes, hw), dtype=np.uint64)
    for i in range(hw):
        planes[:, :, i >> 6] |= bits[:, :, i] << np.uint64(i & 63)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CATTUS_SPLIT_W", mode)  # read by cattus_hip_create
        with HipEvaluator(blob, batch_size=n + 1, plane_words=words, dtype="f16x2") as ev:
            out[mode] = ev.eval(planes)
    assert (out["1"][0] == out["0"][0]).all() and (out["1"][1] == out["0"][1]).all()
    assert np.isfinite(out["1"][0]).all() and np.abs(out["1"][0]).max() > 0


def test_f16x2_range_large_batchnorm_scales():
    """f16 holds 65504 at most.  Weights are safe whatever their size (each output channel is pre-scaled by a power of two
    and un-scaled exactly in the epilogue); activations are stored as they are.  A stem BatchNorm weight of 200 puts
    the residual stream in the hundreds -- the split tower still tracks the f32 oracle to 22 bits --; one of 1e5 sends
    it beyond the f16 range, where the epilogue saturates at 65504 instead of producing infinities -- and COUNTS what it
    clamped: cattus_stats.saturated is 0 for the first network and > 0 for the second, on every kernel that stores f16
    activations (register-ring and LDS-ring split conv, the single-term f16 conv), so a caller can tell that the default
    dtype left its range."""
    from cattus_amd.weights import pack_tensors, seeded_tensors

    d = NetDesc(**CHESS, blocks=2, filters=64, vhc=8, phc=8)
    planes = synth.random_chess_planes(9, 4)
    for scale, finite_only in ((200.0, False), (1e5, True)):
        t = seeded_tensors(d, 6)
        t["_conv1._bn.weight"] = t["_conv1._bn.weight"] * np.float32(scale)
        t["_residual_blocks.0._conv1.weight"] = t["_residual_blocks.0._conv1.weight"] * np.float32(1e-3)  # tiny weights too
        blob = pack_tensors(d, t)
        want_p, want_v = oracle.OracleNet(blob).forward(planes)
        with HipEvaluator(blob, batch_size=16, plane_words=1, dtype="f16x2") as ev:
            assert ev.stats()["saturated"] == 0
            p, v = ev.eval(planes)
            sat = ev.stats()["saturated"]
            ev.eval(planes)
            assert ev.stats()["saturated"] == 2 * sat  # sticky: it accumulates over the evaluator's life
        assert (sat > 0) == finite_only, (scale, sat)
        with HipEvaluator(blob, batch_size=16, plane_words=1, dtype="f16") as ev:
            ev.eval(planes)
            assert (ev.stats()["saturated"] > 0) == finite_only
        with HipEvaluator(blob, batch_size=16, plane_words=1, dtype="f32") as ev:
            ev.eval(planes)
            assert ev.stats()["saturated"] == 0  # the f32 tower has no range to leave
        assert np.isfinite(p).all() and np.isfinite(v).all()
        if not finite_only:
            assert np.abs(want_p).max() > 20  # the scale did reach the logits
            np.testing.assert_allclose(p, want_p, rtol=2e-5, atol=2e-5 * np.abs(want_p).max())
            np.testing.assert_allclose(v, want_v, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["chess_7x16", "hex11_2x8", "ttt_5x8", "hex4_7x16", "hex11_1x1"])
def test_mfma_tower_equals_generic_checker_on_the_reference_shapes(name, monkeypatch):
    """The reference's own nets (7x16, 5x8 filters: training/config/*.yaml) and boards above 8x8
    (training/tests/test_net_output.py:153-189) run on conv3x3_mfma_v2_kernel with zero-padded channels / 128-slot
    boards; the one-thread-per-output kernel stays as a checker (CATTUS_FORCE_GENERIC=1): f32 results are equal
    bit for bit, whatever the batch composition."""
    d, blob, z = blob_for(name)
    planes = z["planes"]
    words = _plane_words(planes)
    rep = np.concatenate([planes] * 3)[: len(planes) * 2 + 1]  # ragged: not a multiple of the boards per workgroup
    monkeypatch.setenv("CATTUS_FORCE_GENERIC", "1")
    with HipEvaluator(blob, batch_size=len(rep), plane_words=words, dtype="f32") as ev:
        want_p, want_v = ev.eval(rep)
    monkeypatch.delenv("CATTUS_FORCE_GENERIC")
    with HipEvaluator(blob, batch_size=len(rep) + 7, plane_words=words, dtype="f32") as ev:
        got_p, got_v = ev.eval(rep)
        one_p, one_v = ev.eval(rep[-1:])
    assert (got_p == want_p).all() and (got_v == want_v).all()
    assert (one_p[0] == want_p[-1]).all() and one_v[0] == want_v[-1]


@pytest.mark.parametrize("game,desc,words,n", [
    ("hex7", dict(**hex_game(7), blocks=6, filters=64, vhc=16, phc=16), 2, 128),    # BASELINE config 2: one board per workgroup
    ("hex7", dict(**hex_game(7), blocks=3, filters=64, vhc=16, phc=16), 2, 1100),   # 128-row workgroups, ragged batch
    ("chess", dict(**CHESS, blocks=7, filters=16, vhc=8, phc=8), 1, 300),           # the reference's chess net, padded channels
    ("hex11", dict(**hex_game(11), blocks=2, filters=8, vhc=4, phc=4), 2, 37),      # 128-slot boards
    ("hex9", dict(**hex_game(9), blocks=2, filters=40, vhc=16, phc=16), 2, 600),    # 128-slot boards, 600 of them
    ("ttt", dict(planes=3, board=3, moves=9, blocks=5, filters=8, vhc=8, phc=8), 1, 5),
    ("ttt", dict(planes=3, board=3, moves=9, blocks=0, filters=8, vhc=8, phc=8), 1, 3),  # a stem and nothing else
])
@pytest.mark.parametrize("dtype", ["bf16", "f16x2"])
def test_resident_tower_equals_per_layer_launches(game, desc, words, n, dtype, monkeypatch):
    """bf16 and f16x2 networks with <= 64 (padded) filters run their whole tower in one launch with the activations
    resident in LDS (tower64_lds_kernel / tower64_split_kernel, plane pack fused); results are bit-identical to the
    per-layer launches (CATTUS_TOWER64=0), for whole and ragged batches and after repeated passes."""
    d = NetDesc(**desc)
    blob = seeded_blob(d, 17)
    rng = np.random.default_rng(5)
    hw = d.board * d.board
    planes = np.zeros((n, d.planes, words), dtype=np.uint64)
    bits = rng.integers(0, 2, size=(n, d.planes, hw), dtype=np.uint64)
    for i in range(hw):
        planes[:, :, i >> 6] |= bits[:, :, i] << np.uint64(i & 63)
    monkeypatch.setenv("CATTUS_TOWER64", "0")
    with HipEvaluator(blob, batch_size=n, plane_words=words, dtype=dtype) as ev:
        want_p, want_v = ev.eval(planes)
        assert ev.time_tower(min(n, 8), 1)[1] == 1 + 2 * d.blocks  # the per-layer launches
    monkeypatch.delenv("CATTUS_TOWER64")
    with HipEvaluator(blob, batch_size=n + 3, plane_words=words, dtype=dtype) as ev:
        for _ in range(3):
            got_p, got_v = ev.eval(planes)
            assert (got_p == want_p).all() and (got_v == want_v).all()
        k = max(1, n // 3)
        sub_p, sub_v = ev.eval(planes[k : 2 * k + 1])
        assert (sub_p == want_p[k : 2 * k + 1]).all() and (sub_v == want_v[k : 2 * k + 1]).all()
        assert ev.time_tower(min(n, 8), 1)[1] == 1  # one tower launch per forward
        assert ev.stats()["saturated"] == 0


def test_resident_tower_row_splits_agree(monkeypatch):
    """The resident tower picks 128- or 64-row workgroups (CH = 2, 4) by batch size, and with 64 rows and a
    board of <= 63 pixels walks a layer as one step; the split changes which wave holds which tile and when the
    waves meet, not the arithmetic of an output element: forced one after the other on the same batch
    (CATTUS_T64_CH / CATTUS_T64_LS, read when an evaluator is created) they give the bits of the per-layer launches."""
    d = NetDesc(**hex_game(7), blocks=4, filters=64, vhc=16, phc=16)
    blob = seeded_blob(d, 23)
    rng = np.random.default_rng(9)
    n = 203
    planes = np.zeros((n, d.planes, 2), dtype=np.uint64)
    bits = rng.integers(0, 2, size=(n, d.planes, 49), dtype=np.uint64)
    for i in range(49):
        planes[:, :, i >> 6] |= bits[:, :, i] << np.uint64(i & 63)
    monkeypatch.setenv("CATTUS_TOWER64", "0")
    with HipEvaluator(blob, batch_size=n, plane_words=2, dtype="bf16") as ev:
        want_p, want_v = ev.eval(planes)
    monkeypatch.delenv("CATTUS_TOWER64")
    for ch, ls in (("4", "1"), ("2", "1"), ("4", "0")):
        monkeypatch.setenv("CATTUS_T64_CH", ch)
        monkeypatch.setenv("CATTUS_T64_LS", ls)  # 64-row workgroups: one barrier per layer (default) or three
        with HipEvaluator(blob, batch_size=n, plane_words=2, dtype="bf16") as ev:
            got_p, got_v = ev.eval(planes)
            assert (got_p == want_p).all() and (got_v == want_v).all(), (ch, ls)
            one_p, one_v = ev.eval(planes[77:78])
            assert (one_p[0] == want_p[77]).all() and one_v[0] == want_v[77], (ch, ls)


def test_resident_split_tower_workgroup_shapes_agree_from_512_boards_up():
    """tower64_split_kernel picks its workgroup shape PER LAUNCH from the batch's row count: two boards per workgroup from 512
    boards up, one below (kernels_t64s.hip).  A leaf's bits must not depend on that: an evaluator of 640 boards (hex7 6x64) under
    every forced shape (CATTUS_T64S_SHAPE=1 / 2 / 9), on a full batch (two-board workgroups by default), a 300-leaf batch (one-board
    workgroups) and a single leaf, against the per-layer launches (CATTUS_TOWER64=0)."""
    d = NetDesc(**hex_game(7), blocks=6, filters=64, vhc=16, phc=16)
    blob = seeded_blob(d, 29)
    n = 640
    planes = synth.random_hex_planes(n, 7, 29)
    with HipEvaluator(blob, batch_size=n, plane_words=2, dtype="f16x2", switches={"CATTUS_TOWER64": "0"}) as ev:
        assert ev.tower_kernel() == "conv3x3_splitw_kernel"
        want_p, want_v = ev.eval(planes)
    for shape in (None, "1", "2", "9"):
        with HipEvaluator(blob, batch_size=n, plane_words=2, dtype="f16x2", switches={"CATTUS_T64S_SHAPE": shape} if shape else {}) as ev:
            assert ev.tower_kernel() == "tower64_split_kernel"
            got_p, got_v = ev.eval(planes)
            assert (got_p == want_p).all() and (got_v == want_v).all(), shape
            part_p, part_v = ev.eval(planes[100:400])  # 300 boards: below the switch
            assert (part_p == want_p[100:400]).all() and (part_v == want_v[100:400]).all(), shape
            one_p, one_v = ev.eval(planes[555:556])
            assert (one_p[0] == want_p[555]).all() and one_v[0] == want_v[555], shape
            assert ev.stats()["saturated"] == 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["chess_20x256", "hex11_2x8"])
def test_fused_stem_equals_separate_plane_pack(name, dtype, monkeypatch):
    """The stem conv of the per-layer path expands the bitboard planes in its loader waves (K0 fused); with
    CATTUS_FUSED_STEM=0 the planes go through pack_planes_nhwc_kernel first.  Same bits either way."""
    d, blob, z = blob_for(name)
    planes = np.concatenate([z["planes"]] * 2)[: len(z["planes"]) + 3]
    words = _plane_words(planes)
    monkeypatch.setenv("CATTUS_TOWER64", "0")  # the per-layer path also for the small net
    monkeypatch.setenv("CATTUS_FUSED_STEM", "0")
    with HipEvaluator(blob, batch_size=len(planes), plane_words=words, dtype=dtype) as ev:
        want_p, want_v = ev.eval(planes)
    monkeypatch.delenv("CATTUS_FUSED_STEM")
    with HipEvaluator(blob, batch_size=len(planes) + 2, plane_words=words, dtype=dtype) as ev:
        got_p, got_v = ev.eval(planes)
    assert (got_p == want_p).all() and (got_v == want_v).all()


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16x2"])
@pytest.mark.parametrize("game,desc,words,n", [
    ("chess", dict(**CHESS, blocks=3, filters=256, vhc=8, phc=8), 1, 70),
    ("hex11", dict(**hex_game(11), blocks=2, filters=128, vhc=16, phc=16), 2, 21),
])
def test_half_cout_workgroups_equal_full_ones(game, desc, words, n, dtype, monkeypatch):
    """Small batches run the conv layers on 256-row x 32-cout workgroups (twice as many) so that more of the
    chip is busy; same MFMA shape and k order, so the bits do not depend on which tile ran."""
    d = NetDesc(**desc)
    blob = seeded_blob(d, 23)
    rng = np.random.default_rng(11)
    hw = d.board * d.board
    planes = np.zeros((n, d.planes, words), dtype=np.uint64)
    bits = rng.integers(0, 2, size=(n, d.planes, hw), dtype=np.uint64)
    for i in range(hw):
        planes[:, :, i >> 6] |= bits[:, :, i] << np.uint64(i & 63)
    outs = []
    for cb in ("2", "1"):
        monkeypatch.setenv("CATTUS_CONV_CB", cb)
        with HipEvaluator(blob, batch_size=n, plane_words=words, dtype=dtype) as ev:
            outs.append(ev.eval(planes))
    monkeypatch.delenv("CATTUS_CONV_CB")
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
    if dtype == "f32":
        want_p, want_v = oracle.OracleNet(blob).forward(planes[:8])
        assert (outs[1][0][:8] == want_p).all() and (outs[1][1][:8] == want_v).all()


@pytest.mark.parametrize("game,desc,words,n", [
    ("chess", dict(**CHESS, blocks=3, filters=256, vhc=8, phc=8), 1, 40),            # 64-slot boards: two waves per board
    ("hex11", dict(**hex_game(11), blocks=2, filters=128, vhc=16, phc=16), 2, 21),   # 128-slot boards: one board per workgroup
    ("ttt", dict(planes=3, board=3, moves=9, blocks=2, filters=64, vhc=8, phc=8), 1, 7),
])
@pytest.mark.parametrize("dtype", ["f16x2", "f32", "bf16"])
def test_half_row_workgroups_equal_full_ones(game, desc, words, n, dtype, monkeypatch):
    """Batches that would leave half of the CUs empty run the conv layers on 128-row x 32-cout workgroups (32 pixels per
    consumer wave); CATTUS_CONV_PBW=2 keeps the 256-row ones, =1 forces the small ones wherever the tile is 32 couts: same
    MFMA sequence per output, same bits (and, in f32, the oracle's)."""
    d = NetDesc(**desc)
    blob = seeded_blob(d, 29)
    rng = np.random.default_rng(13)
    hw = d.board * d.board
    planes = np.zeros((n, d.planes, words), dtype=np.uint64)
    bits = rng.integers(0, 2, size=(n, d.planes, hw), dtype=np.uint64)
    for i in range(hw):
        planes[:, :, i >> 6] |= bits[:, :, i] << np.uint64(i & 63)
    outs = []
    for pbw in ("2", "1", None):
        if pbw is None:
            monkeypatch.delenv("CATTUS_CONV_PBW")
        else:
            monkeypatch.setenv("CATTUS_CONV_PBW", pbw)
        with HipEvaluator(blob, batch_size=n, plane_words=words, dtype=dtype) as ev:
            outs.append(ev.eval(planes))
    for o in outs[1:]:
        assert (o[0] == outs[0][0]).all() and (o[1] == outs[0][1]).all()
    if dtype == "f32":
        want_p, want_v = oracle.OracleNet(blob).forward(planes[:6])
        assert (outs[2][0][:6] == want_p).all() and (outs[2][1][:6] == want_v).all()


def test_wide_heads_take_the_generic_path_and_refuse_bf16():
    d = NetDesc(**hex_game(5), blocks=1, filters=32, vhc=24, phc=24)  # 48 head channels > one 32-row MFMA tile
    blob = seeded_blob(d, 4)
    planes = synth.random_hex_planes(5, 5, 2)
    want_p, want_v = oracle.OracleNet(blob).forward(planes)
    with HipEvaluator(blob, batch_size=8, plane_words=2, dtype="f32") as ev:
        got_p, got_v = ev.eval(planes)
    assert (got_p == want_p).all() and (got_v == want_v).all()
    with pytest.raises(CattusHipError) as ei:
        HipEvaluator(blob, batch_size=4, plane_words=2, dtype="bf16")
    assert ei.value.status == -2


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16x2"])
def test_rows_independent_of_batch_composition(dtype):
    d = NetDesc(**CHESS, blocks=3, filters=64, vhc=8, phc=8)
    blob = seeded_blob(d, 77)
    planes = synth.random_chess_planes(37, 5)
    with HipEvaluator(blob, batch_size=64, plane_words=1, dtype=dtype) as ev:
        p_all, v_all = ev.eval(planes)
        # one at a time, reversed order, and a ragged sub-batch
        for i in (0, 1, 17, 36):
            p1, v1 = ev.eval(planes[i : i + 1])
            assert (p1[0] == p_all[i]).all() and v1[0] == v_all[i]
        p_rev, v_rev = ev.eval(planes[::-1].copy())
        assert (p_rev[::-1] == p_all).all() and (v_rev[::-1] == v_all).all()
        p_sub, v_sub = ev.eval(planes[5:30])
        assert (p_sub == p_all[5:30]).all() and (v_sub == v_all[5:30]).all()
        # determinism across repeats (reference repeats 8x: test_net_output.py:21-22)
        for _ in range(8):
            p2, v2 = ev.eval(planes)
            assert (p2 == p_all).all() and (v2 == v_all).all()


def test_eval_rejects_bad_sample_len():
    d, blob, z = blob_for("ttt_2x64")
    with HipEvaluator(blob, batch_size=4, plane_words=1, dtype="f32") as ev:
        with pytest.raises(CattusHipError):
            ev.eval(np.concatenate([z["planes"], z["planes"]])[:5])  # n > batch_size
        with pytest.raises(CattusHipError):
            ev.eval(z["planes"][:0])  # n == 0


def test_non_finite_logits_are_scrubbed():
    # engine/src/net/mod.rs:56-61: !is_finite -> f32::MIN
    d = NetDesc(**hex_game(4), blocks=1, filters=64, vhc=4, phc=4)
    from cattus_amd.weights import pack_tensors, seeded_tensors

    t = seeded_tensors(d, 5)
    t["_policy_head.2.bias"][3] = np.inf
    t["_policy_head.2.bias"][7] = np.nan
    blob = pack_tensors(d, t)
    planes = synth.random_hex_planes(3, 4, 9)
    want_p, want_v = oracle.OracleNet(blob).forward(planes)
    for dtype in ("f32", "bf16", "f16x2"):
        with HipEvaluator(blob, batch_size=4, plane_words=2, dtype=dtype) as ev:
            p, v = ev.eval(planes)
        fmin = np.finfo(np.float32).min
        assert (p[:, 3] == fmin).all() and (p[:, 7] == fmin).all()
        assert np.isfinite(p).all()
    assert (want_p[:, 3] == fmin).all() and (want_p[:, 7] == fmin).all()


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16x2"])
def test_leaf_server_matches_blocking_eval(dtype):
    d = NetDesc(**hex_game(7), blocks=2, filters=64, vhc=16, phc=16)
    blob = seeded_blob(d, 3)
    planes = synth.random_hex_planes(50, 7, 4)
    with HipEvaluator(blob, batch_size=16, plane_words=2, dtype=dtype, flush_us=2000) as ev:
        want_p, want_v = ev.eval(planes[:16])
        want_p2, want_v2 = ev.eval(planes[16:32])
        want_p3, want_v3 = ev.eval(planes[32:48])
        want_p4, want_v4 = ev.eval(planes[48:50])
        want_p = np.concatenate([want_p, want_p2, want_p3, want_p4])
        want_v = np.concatenate([want_v, want_v2, want_v3, want_v4])
        results = [None] * len(planes)

        def worker(ids):
            for i in ids:
                t = ev.submit(planes[i])
                results[i] = ev.wait(t)

        threads = [threading.Thread(target=worker, args=(range(k, len(planes), 10),)) for k in range(10)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for i, (p, v) in enumerate(results):
            assert (p == want_p[i]).all() and v == want_v[i]
        st = ev.stats()
        assert st["positions"] >= 100 and st["batches"] >= 8
        # a ticket can be collected once
        t = ev.submit(planes[0])
        ev.flush()
        ev.wait(t)
        with pytest.raises(CattusHipError):
            ev.wait(t)


def test_full_size_chess_20x256_batch_256():
    """BASELINE config 3 at full size: f32 is bit-exact against the oracle on a sample of the rows;
    all 256 rows are checked through size-independent properties (permutation equivariance,
    ragged-batch agreement) in both dtypes; bf16 stays within its tolerance of f32 on every row."""
    d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
    blob = seeded_blob(d, 2)
    planes = synth.random_chess_planes(256, 2)
    perm = np.random.default_rng(0).permutation(256)
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f32") as ev:
        p32, v32 = ev.eval(planes)
        pp, vp = ev.eval(planes[perm])
        assert (pp == p32[perm]).all() and (vp == v32[perm]).all()
        pr, vr = ev.eval(planes[:255])
        assert (pr == p32[:255]).all() and (vr == v32[:255]).all()
    rows = np.arange(0, 256, 16)
    want_p, want_v = oracle.OracleNet(blob).forward(planes[rows])
    assert (p32[rows] == want_p).all() and (v32[rows] == want_v).all()
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2") as ev:
        ps, vs = ev.eval(planes)
        pp, vp = ev.eval(planes[perm])
        assert (pp == ps[perm]).all() and (vp == vs[perm]).all()
        pr, vr = ev.eval(planes[:255])
        assert (pr == ps[:255]).all() and (vr == vs[:255]).all()
    # the split tower against the bit-exact f32 tower, all 256 rows: inside the reference's cross-runtime bar, and two
    # f32-grade roundings apart at most (each is <= 1e-6 / 3e-7 from the float64 truth)
    assert outputs_equal_ref_tol(ps, vs, p32, v32)
    assert np.abs(ps - p32).max() <= 3e-6 and np.abs(vs - v32).max() <= 1e-6, (np.abs(ps - p32).max(), np.abs(vs - v32).max())
    assert (ps.argmax(1) == p32.argmax(1)).all()
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="bf16") as ev:
        p16, v16 = ev.eval(planes)
        pp, vp = ev.eval(planes[perm])
        assert (pp == p16[perm]).all() and (vp == v16[perm]).all()
    print("chess 20x256, 256 leaves: bf16 vs f32 max |dlogit| %.4g |dvalue| %.4g; f16x2 vs f32 %.3g %.3g" % (
        np.abs(p16 - p32).max(), np.abs(v16 - v32).max(), np.abs(ps - p32).max(), np.abs(vs - v32).max()))
    assert np.abs(p16 - p32).max() <= BF16_FULL["20x256"][0], np.abs(p16 - p32).max()
    assert np.abs(v16 - v32).max() <= BF16_FULL["20x256"][1], np.abs(v16 - v32).max()
    # greedy move agreement between the two dtypes (reported, loosely bounded)
    agree = (p16.argmax(1) == p32.argmax(1)).mean()
    assert agree >= 0.9, agree


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16x2"])
def test_eval_legal_softmax_bit_exact_vs_restatement(dtype):
    """cattus_hip_eval_legal = cattus_hip_eval + calc_moves_probs (net/mod.rs:100-119) on the device:
    bit-exact against the oracle's restatement applied to the same evaluator's logits, within 1e-6 of
    the libm softmax, zero past each leaf's count; ragged counts including 1 and the full stride."""
    d = NetDesc(**CHESS, blocks=2, filters=64, vhc=8, phc=8)
    blob = seeded_blob(d, 21)
    n, L = 37, 224
    planes = synth.random_chess_planes(n, 3)
    rng = np.random.default_rng(1)
    cnt = rng.integers(1, L + 1, size=n).astype(np.uint16)
    cnt[0], cnt[1], cnt[2] = 1, L, 2
    idx = np.zeros((n, L), dtype=np.uint16)
    for i in range(n):
        idx[i, : cnt[i]] = rng.choice(d.moves, cnt[i], replace=False)
    with HipEvaluator(blob, batch_size=64, plane_words=1, dtype=dtype) as ev:
        pol, val = ev.eval(planes)
        probs, val2 = ev.eval_legal(planes, idx, cnt)
        assert (val == val2).all()
        for i in range(n):
            c = int(cnt[i])
            want = oracle.softmax_legal_det(pol[i], idx[i, :c])
            assert (probs[i, :c] == want).all(), (i, np.abs(probs[i, :c] - want).max())
            assert (probs[i, c:] == 0).all()
            libm = oracle.softmax_legal(pol[i], idx[i, :c].astype(np.uint32))
            assert np.abs(probs[i, :c] - libm).max() <= 1e-6
        # argument checks
        bad = idx.copy()
        bad[3, 0] = d.moves
        with pytest.raises(CattusHipError):
            ev.eval_legal(planes, bad, cnt)
        with pytest.raises(CattusHipError):
            ev.eval_legal(planes, idx[:, :100], cnt)  # count > stride


def test_eval_legal_scrubbed_logits_get_zero_probability():
    # a non-finite logit is scrubbed to f32::MIN (net/mod.rs:56-61); its softmax weight is exactly 0
    from cattus_amd.weights import pack_tensors, seeded_tensors

    d = NetDesc(**hex_game(4), blocks=1, filters=64, vhc=4, phc=4)
    t = seeded_tensors(d, 5)
    t["_policy_head.2.bias"][3] = np.inf
    blob = pack_tensors(d, t)
    planes = synth.random_hex_planes(2, 4, 9)
    idx = np.tile(np.arange(16, dtype=np.uint16), (2, 1))
    cnt = np.array([16, 5], dtype=np.uint16)
    with HipEvaluator(blob, batch_size=4, plane_words=2, dtype="f32") as ev:
        probs, _ = ev.eval_legal(planes, idx, cnt)
    assert (probs[:, 3] == 0).all()
    assert abs(probs[0].sum() - 1) < 1e-5 and abs(probs[1, :5].sum() - 1) < 1e-5 and (probs[1, 5:] == 0).all()


def test_full_size_chess_40x384_batch_512():
    """BASELINE config 5 at full size (107 M parameters, 13.6 GFLOP per leaf, batch 512): f32 bit-exact against
    the oracle on a few rows; every row through size-independent properties (permutation equivariance, ragged
    sub-batch agreement); bf16 within its stated tolerance of f32 on every row."""
    d = NetDesc(**CHESS, blocks=40, filters=384, vhc=8, phc=8)
    blob = seeded_blob(d, 5)
    planes = synth.random_chess_planes(512, 5)
    perm = np.random.default_rng(1).permutation(512)
    with HipEvaluator(blob, batch_size=512, plane_words=1, dtype="f32") as ev:
        p32, v32 = ev.eval(planes)
        pp, vp = ev.eval(planes[perm])
        assert (pp == p32[perm]).all() and (vp == v32[perm]).all()
        pr, vr = ev.eval(planes[100:357])
        assert (pr == p32[100:357]).all() and (vr == v32[100:357]).all()
    rows = np.array([0, 255, 511])
    want_p, want_v = oracle.OracleNet(blob).forward(planes[rows])
    assert (p32[rows] == want_p).all() and (v32[rows] == want_v).all()
    with HipEvaluator(blob, batch_size=512, plane_words=1, dtype="bf16") as ev:
        p16, v16 = ev.eval(planes)
        pp, vp = ev.eval(planes[perm])
        assert (pp == p16[perm]).all() and (vp == v16[perm]).all()
    assert np.isfinite(p16).all() and np.isfinite(v16).all()
    print("chess 40x384, 512 leaves: bf16 vs f32 max |dlogit| %.4g |dvalue| %.4g" % (np.abs(p16 - p32).max(), np.abs(v16 - v32).max()))
    assert np.abs(p16 - p32).max() <= BF16_FULL["40x384"][0], np.abs(p16 - p32).max()
    assert np.abs(v16 - v32).max() <= BF16_FULL["40x384"][1], np.abs(v16 - v32).max()
    with HipEvaluator(blob, batch_size=512, plane_words=1, dtype="f16x2") as ev:
        ps, vs = ev.eval(planes)
        pr, vr = ev.eval(planes[100:357])
        assert (pr == ps[100:357]).all() and (vr == vs[100:357]).all()
    print("chess 40x384, 512 leaves: f16x2 vs f32 max |dlogit| %.3g |dvalue| %.3g" % (np.abs(ps - p32).max(), np.abs(vs - v32).max()))
    # 81 layers deep, two f32-grade computations of the same network differ by a few 1e-6 at the logits (measured 5.8e-6 /
    # 9.9e-7 against the bit-exact f32 tower): bounded at twice that; the reference's own bar is stated for nets of <= 7 blocks
    assert np.abs(ps - p32).max() <= 1.2e-5 and np.abs(vs - v32).max() <= 2e-6, (np.abs(ps - p32).max(), np.abs(vs - v32).max())
    assert (ps.argmax(1) == p32.argmax(1)).all()


def test_device_pointer_entry_points_and_lanes_agree_with_host_entry_point():
    """cattus_hip_eval_device / _lane (HBM-resident buffers, asynchronous on a caller's stream; what bench.py
    times) give bit for bit what the blocking host-buffer entry point gives, on either lane and with both
    lanes in flight on two streams."""
    torch = pytest.importorskip("torch")
    d = NetDesc(**CHESS, blocks=3, filters=64, vhc=8, phc=8)
    blob = seeded_blob(d, 8)
    planes = synth.random_chess_planes(96, 8)
    dev = torch.device("cuda", 0)
    with HipEvaluator(blob, batch_size=128, plane_words=1, dtype="bf16") as ev:
        want_p, want_v = ev.eval(planes)
        d_planes = torch.from_numpy(planes.view(np.int64)).to(dev)
        outs = []
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        for rep in range(3):
            for lane in (0, 1):
                pol = torch.empty((96, d.moves), dtype=torch.float32, device=dev)
                val = torch.empty((96,), dtype=torch.float32, device=dev)
                ev.eval_device(d_planes.data_ptr(), 96, pol.data_ptr(), val.data_ptr(), streams[lane].cuda_stream, lane=lane)
                outs.append((pol, val))
        torch.cuda.synchronize()
        for pol, val in outs:
            assert (pol.cpu().numpy() == want_p).all() and (val.cpu().numpy() == want_v).all()
        with pytest.raises(CattusHipError):
            ev.eval_device(d_planes.data_ptr(), 96, outs[0][0].data_ptr(), outs[0][1].data_ptr(), 0, lane=2)
        # Stream ordering alone is enough (no device-wide synchronize): the work runs on the caller's stream,
        # also when that is the legacy default stream (handle 0), and on the lane's own stream.
        pol = torch.zeros((96, d.moves), dtype=torch.float32, device=dev)
        val = torch.zeros((96,), dtype=torch.float32, device=dev)
        s = streams[0]
        ev.eval_device(d_planes.data_ptr(), 96, pol.data_ptr(), val.data_ptr(), s.cuda_stream, lane=0)
        with torch.cuda.stream(s):
            got = pol.clone()
        s.synchronize()
        assert (got.cpu().numpy() == want_p).all()
        pol.zero_()
        torch.cuda.synchronize()
        ev.eval_device(d_planes.data_ptr(), 96, pol.data_ptr(), val.data_ptr(), 0, lane=1)
        got = pol.clone()  # torch's current stream is the default stream: ordered behind the evaluation
        torch.cuda.default_stream(dev).synchronize()
        assert (got.cpu().numpy() == want_p).all()
        assert ev.lane_stream(0) != 0 and ev.lane_stream(0) != ev.lane_stream(1)


# ---- the split tower in Winograd F(2x2, 3x3) form: f16x2 evaluators of 8x8-board networks with a multiple of 64 filters (>= 128) and
# max_batch > 128 (or tower_form "winograd").  Two kernels, same bits: k4 = conv3x3_wino4_kernel (kernels_wino4.hip: 4 frequencies x
# 2x2 blocks per wave, the default wherever it covers the shape), k16 = conv3x3_wino_kernel (kernels_wino.hip: 16 frequencies of one block) ----
WINO_POLICY_ATOL_VS_F64, WINO_VALUE_ATOL_VS_F64 = 1.5e-6, 5e-7  # measured 5.1e-7 / 1.3e-7 (the direct split tower: 6.3e-7 / 2.1e-7)
# k4 runs as ONE launch (tower_wino4_kernel: the layers chained by hand-off counters) while its grid fits the device, else per layer
WINO_KERNELS = {"k4": ("tower_wino4_kernel", "conv3x3_wino4_kernel"), "k16": ("conv3x3_wino_kernel",), "k8": ("conv3x3_wino8_kernel",)}


def wino_eval(blob, batch_size, wk, **more):
    return HipEvaluator(blob, batch_size=batch_size, plane_words=1, dtype="f16x2", tower_form="winograd", switches={"CATTUS_WINO_KERNEL": wk, **more})


@pytest.mark.parametrize("wk", ["k4", "k16"])
def test_winograd_tower_within_the_reference_tolerance_and_batch_independent(wk):
    """chess 20x256 (the reference-made fixture): with max_batch > 128 the f16x2 tower runs its 40 layers behind the stem in
    Winograd form -- 2.25x fewer MFMAs, operands transformed in f32 / float64 and split into f16 pairs, f32 activations between the
    layers.  Inside the reference's cross-runtime bar (training/tests/test_net_output.py:28-33), as close to the reference's float64
    run as the direct split tower, the same bits whatever the batch a leaf comes in, and selected by the configuration alone."""
    d, blob, z = blob_for("chess_20x256")
    planes = z["planes"]
    rep = np.concatenate([planes] * 50)[:197]
    with wino_eval(blob, 256, wk) as ev:
        assert ev.tower_kernel() in WINO_KERNELS[wk]
        p, v = ev.eval(planes)
        pr, vr = ev.eval(rep)
        assert ev.stats()["saturated"] == 0
    assert outputs_equal_ref_tol(p, v, z["policy"], z["value"])
    assert np.abs(p - z["policy_f64"]).max() <= WINO_POLICY_ATOL_VS_F64 and np.abs(v - z["value_f64"]).max() <= WINO_VALUE_ATOL_VS_F64
    for i in range(len(rep)):  # a leaf's bits do not depend on the batch or the slot
        assert (pr[i] == p[i % len(planes)]).all() and vr[i] == v[i % len(planes)]
    with HipEvaluator(blob, batch_size=128, plane_words=1, dtype="f16x2", switches={}) as ev:
        assert ev.tower_kernel() == "conv3x3_splitw_kernel"  # small batches: the direct kernels' small tiles
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2", switches={}) as ev:
        assert ev.tower_kernel() == "tower_wino4_kernel"  # what cattus_hip_create chooses by itself
    with HipEvaluator(blob, batch_size=16, plane_words=1, dtype="f16x2", tower_form="winograd", switches={"CATTUS_WINO_KERNEL": wk}) as ev:
        assert ev.tower_kernel() in WINO_KERNELS[wk]  # the form is a field of the configuration: a 16-leaf evaluator in Winograd form ...
        ps, vs = ev.eval(planes)
    assert (ps == p).all() and (vs == v).all()  # ... gives a leaf the bits the 256-leaf evaluator gives it
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2", tower_form="direct", switches={}) as ev:
        assert ev.tower_kernel() == "conv3x3_splitw_kernel"
        pd, vd = ev.eval(planes)
    assert np.abs(p - pd).max() < 2e-6 and np.abs(v - vd).max() < 1e-6  # the two forms of the same tower


def test_winograd_tower_form_is_refused_where_no_kernel_covers_the_shape():
    from cattus_amd.evaluator import CattusHipError

    d = NetDesc(**hex_game(7), blocks=2, filters=128, vhc=16, phc=16)  # a 7x7 board
    with pytest.raises(CattusHipError) as ei:
        HipEvaluator(seeded_blob(d, 1), batch_size=16, plane_words=2, dtype="f16x2", tower_form="winograd", switches={})
    assert ei.value.status == -2  # CATTUS_E_UNSUPPORTED


def test_auto_form_follows_max_batch_and_never_the_batch():
    """cattus_eval_config.tower_form = AUTO: the direct kernels up to max_batch 128, the Winograd tower above (scripts/by_batch_forms.py:
    0.97 | 0.90 ms per step at 128 leaves, 1.45 | 0.93 at 160) -- fixed when the evaluator is created: a 3-leaf batch of a 256-leaf
    evaluator runs the kernel its full batches run, and its leaves get the bits they get in a full batch."""
    d = NetDesc(**CHESS, blocks=2, filters=128, vhc=8, phc=8)
    blob = seeded_blob(d, 12)
    planes = synth.random_chess_planes(160, 12)
    with HipEvaluator(blob, batch_size=128, plane_words=1, dtype="f16x2", switches={}) as ev:
        assert ev.tower_kernel() == "conv3x3_splitw_kernel"
    with HipEvaluator(blob, batch_size=129, plane_words=1, dtype="f16x2", switches={}) as ev:
        assert ev.tower_kernel() == "tower_wino4_kernel"
    with HipEvaluator(blob, batch_size=160, plane_words=1, dtype="f16x2", switches={}) as ev:
        assert ev.tower_kernel() == "tower_wino4_kernel"
        full = ev.eval(planes)
        few = ev.eval(planes[:3])
        assert ev.tower_kernel() == "tower_wino4_kernel"
    assert (few[0] == full[0][:3]).all() and (few[1] == full[1][:3]).all()


@pytest.mark.parametrize("shape", [(3, 128, 256), (2, 192, 200), (2, 256, 24), (1, 384, 512)])
def test_winograd_kernels_agree_bit_for_bit(shape):
    """conv3x3_wino4_kernel and conv3x3_wino8_kernel against conv3x3_wino_kernel: per accumulator the same MFMA sequence, V and Y combined in the same order --
    the same bits, on full, ragged and multi-round grids (192 filters: the 4-frequency kernel alone covers them, checked against the
    direct form's tolerance instead)."""
    blocks, filters, n = shape
    d = NetDesc(**CHESS, blocks=blocks, filters=filters, vhc=8, phc=8)
    blob = seeded_blob(d, 21)
    planes = synth.random_chess_planes(n, 13)
    with wino_eval(blob, max(n, 192), "k4", CATTUS_WINO_PERSIST="0") as ev:
        assert ev.tower_kernel() == "conv3x3_wino4_kernel"
        a = ev.eval(planes)
        a2 = ev.eval(planes[: n // 3 + 1])
        assert ev.stats()["saturated"] == 0
    assert (a2[0] == a[0][: n // 3 + 1]).all() and (a2[1] == a[1][: n // 3 + 1]).all()
    with wino_eval(blob, max(n, 192), "k8") as ev:  # the eight-wave kernel (two waves per SIMD; built, measured, not the default)
        assert ev.tower_kernel() == "conv3x3_wino8_kernel"
        c8 = ev.eval(planes)
        assert ev.stats()["saturated"] == 0
    assert (a[0] == c8[0]).all() and (a[1] == c8[1]).all()
    if filters % 128 == 0:
        with wino_eval(blob, max(n, 192), "k16") as ev:
            assert ev.tower_kernel() == "conv3x3_wino_kernel"
            b = ev.eval(planes)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    else:
        with HipEvaluator(blob, batch_size=max(n, 192), plane_words=1, dtype="f16x2", tower_form="direct", switches={}) as ev:
            b = ev.eval(planes)
        assert np.abs(a[0] - b[0]).max() < 2e-6 and np.abs(a[1] - b[1]).max() < 1e-6


@pytest.mark.parametrize("shape", [(3, 128, 256), (3, 128, 512), (20, 256, 256), (2, 384, 168), (2, 192, 77), (2, 384, 512), (3, 256, 384)])
def test_one_launch_winograd_tower_equals_the_per_layer_launches(shape):
    """tower_wino4_kernel -- every layer behind the stem in one launch, a workgroup keeping its (4 boards x 64 couts) tile through all
    of them, the layers chained by a counter per (layer, board group) -- against the same layers as launches of their own
    (CATTUS_WINO_PERSIST=0): the same bits, on a full grid (one workgroup per CU), on partial batches of the same evaluator, on both
    lanes back to back, and again after many passes (stale lines in a CU's L1 or a missed hand-off would show as wrong rows).  The last
    two shapes have more tiles than the device has CUs (768: three per workgroup and layer -- config 5's; 384: the second round half
    empty): a workgroup walks its tiles layer by layer, board groups straddle the rounds."""
    blocks, filters, n = shape
    d = NetDesc(**CHESS, blocks=blocks, filters=filters, vhc=8, phc=8)
    blob = seeded_blob(d, 31)
    planes = synth.random_chess_planes(n, 17)
    with wino_eval(blob, n, "k4", CATTUS_WINO_PERSIST="0") as ev:
        assert ev.tower_kernel() == "conv3x3_wino4_kernel"
        want = ev.eval(planes)
    with wino_eval(blob, n, "k4") as ev:
        assert ev.tower_kernel() == "tower_wino4_kernel"  # (n / 4 board groups) x (filters / 64 cout groups) tiles, on 256 CUs
        for rep in range(6):
            got = ev.eval(planes)  # alternates between the lanes' buffers
            assert (got[0] == want[0]).all() and (got[1] == want[1]).all(), rep
            k = 1 + (rep * 37) % n
            part = ev.eval(planes[:k])
            assert (part[0] == want[0][:k]).all() and (part[1] == want[1][:k]).all(), (rep, k)
        assert ev.stats()["saturated"] == 0
        assert ev.tower_kernel() == "tower_wino4_kernel"  # no hand-off wait gave up


def test_one_launch_tower_whose_hand_off_gives_up_is_run_again_per_layer():
    """The failure path of the one-launch tower: with a budget of ONE poll per hand-off wait (CATTUS_WINO_SPIN=1) some workgroup finds
    its producers not yet counted in, raises the launch's error word and goes on on rows that are not ready; the host finds the word
    behind the batch, throws the outputs away, runs the batch on the per-layer launches and stays there.  The caller sees the right
    bits and an evaluator that says which kernel it now runs."""
    d = NetDesc(**CHESS, blocks=6, filters=256, vhc=8, phc=8)
    blob = seeded_blob(d, 33)
    planes = synth.random_chess_planes(256, 19)
    with wino_eval(blob, 256, "k4", CATTUS_WINO_PERSIST="0") as ev:
        want = ev.eval(planes)
    with wino_eval(blob, 256, "k4", CATTUS_WINO_SPIN="1") as ev:
        assert ev.tower_kernel() == "tower_wino4_kernel"
        for _ in range(3):
            got = ev.eval(planes)
            assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
        assert ev.tower_kernel() == "conv3x3_wino4_kernel"  # a wait gave up somewhere in 12 layers x 256 workgroups x 3 passes


@pytest.mark.parametrize("wk", ["k4", "k16"])
def test_winograd_tower_on_the_reference_made_fixture_with_the_reference_positions(wk):
    """tests/golden/chess_4x128.npz: a Winograd-shaped network (8x8 board, 128 filters) evaluated by the reference's own module
    (oracle/gen_golden.py) on the reference's five test FENs (training/tests/test_net_output.py:198-204) and 59 synthetic leaves.
    Every leaf inside the reference's cross-runtime tolerance (test_net_output.py:28-33) and within the split tower's distance of
    the reference's float64 run."""
    d, blob, z = blob_for("chess_4x128")
    planes = z["planes"]
    assert len(planes) == 64
    with wino_eval(blob, 256, wk) as ev:
        assert ev.tower_kernel() in WINO_KERNELS[wk]
        p, v = ev.eval(planes)
        assert ev.stats()["saturated"] == 0
    assert outputs_equal_ref_tol(p, v, z["policy"], z["value"])
    assert np.abs(p - z["policy_f64"]).max() <= WINO_POLICY_ATOL_VS_F64 and np.abs(v - z["value_f64"]).max() <= WINO_VALUE_ATOL_VS_F64


@pytest.mark.parametrize("wk", ["k4", "k16"])
def test_winograd_tower_memory_plans_agree_bit_for_bit(wk):
    """The Winograd tower carves U of all layers and its activation buffers from one contiguous allocation (Infinity Cache page
    colouring) and writes a residual block's output over its own skip rows (two activation buffers per lane).
    Neither may change a bit: separate allocations (CATTUS_ARENA=0) and a third buffer (CATTUS_WINO_INPLACE=0) give the same
    outputs, on both lanes, for a full and for a ragged batch."""
    d = NetDesc(**CHESS, blocks=3, filters=128, vhc=8, phc=8)
    blob = seeded_blob(d, 11)
    planes = synth.random_chess_planes(256, 12)
    got = {}
    for arena, inplace in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        with wino_eval(blob, 256, wk, CATTUS_ARENA=arena, CATTUS_WINO_INPLACE=inplace) as ev:
            assert ev.tower_kernel() in WINO_KERNELS[wk]
            a = ev.eval(planes)
            b = ev.eval(planes[:77])  # the second call takes the other lane's buffers when the first one's are still warm
            c = ev.eval(planes)
            assert ev.stats()["saturated"] == 0
        assert (a[0] == c[0]).all() and (a[1] == c[1]).all() and (b[0] == a[0][:77]).all() and (b[1] == a[1][:77]).all()
        got[arena, inplace] = a
    ref = got["1", "1"]
    for k, v in got.items():
        assert (v[0] == ref[0]).all() and (v[1] == ref[1]).all(), k


@pytest.mark.parametrize("net,n,bound", [("20x256", 256, (3e-6, 1e-6)), ("40x384", 512, (1.2e-5, 2e-6))])
def test_winograd_tower_tracks_the_f32_tower_at_full_size(net, n, bound):
    """BASELINE configs 3 and 5 at full size in Winograd form against the bit-exact f32 tower on EVERY leaf of the batch, under the
    bounds the direct split tower is held to (measured on 256 / 512 leaves: 1.1e-6 / 3.1e-7 and 3.5e-6 / 8e-7)."""
    blocks, filters = (20, 256) if net == "20x256" else (40, 384)
    d = NetDesc(**CHESS, blocks=blocks, filters=filters, vhc=8, phc=8)
    blob = seeded_blob(d, 2 if net == "20x256" else 3)
    planes = synth.random_chess_planes(n, 2 if net == "20x256" else 3)
    with HipEvaluator(blob, batch_size=n, plane_words=1, dtype="f32") as ev:
        want_p, want_v = ev.eval(planes)
    with HipEvaluator(blob, batch_size=n, plane_words=1, dtype="f16x2", switches={}) as ev:
        assert ev.tower_kernel() == "tower_wino4_kernel"  # what bench.py's headline is timed on (40x384 at 512 leaves: 768 tiles, three per workgroup and layer)
        got_p, got_v = ev.eval(planes)
    assert np.isfinite(got_p).all() and np.isfinite(got_v).all()
    assert np.abs(got_p - want_p).max() <= bound[0] and np.abs(got_v - want_v).max() <= bound[1]


@pytest.mark.parametrize("wk", ["k4", "k16"])
def test_winograd_tower_counts_inputs_that_leave_the_f16_range(wk):
    """The Winograd tower keeps f32 activations; what leaves the f16 range there is a TRANSFORMED input (up to four times an
    activation), so activations are capped at 65504 / 4 where they are written and counted like the direct tower's: 0 for a
    BatchNorm scale of 200, > 0 for 1e5."""
    from cattus_amd.weights import pack_tensors, seeded_tensors

    d = NetDesc(**CHESS, blocks=2, filters=128, vhc=8, phc=8)
    planes = synth.random_chess_planes(9, 4)
    for scale, saturates in ((200.0, False), (1e5, True)):
        t = seeded_tensors(d, 6)
        t["_conv1._bn.weight"] = t["_conv1._bn.weight"] * np.float32(scale)
        blob = pack_tensors(d, t)
        with wino_eval(blob, 256, wk) as ev:
            assert ev.tower_kernel() in WINO_KERNELS[wk]
            p, v = ev.eval(planes)
            assert (ev.stats()["saturated"] > 0) == saturates
        assert np.isfinite(p).all() and np.isfinite(v).all()
        if not saturates:
            want_p, want_v = oracle.OracleNet(blob).forward(planes)
            np.testing.assert_allclose(p, want_p, rtol=2e-5, atol=2e-5 * np.abs(want_p).max())
