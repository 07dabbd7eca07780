"""The reference's other ``model.type``: ``SimpleTwoHeadedModel`` (training/cattus_train/net_utils.py:92-121, selected
at train_process.py:377-382; the TTT/hex test configs use it).  Fixtures under tests/golden/simple/ are outputs of the
reference's own module (oracle/gen_golden.py).  CPU tests pin the oracle and the weight blob; the GPU tests compare the
HIP evaluator with the oracle, bit for bit, through the C ABI."""

import numpy as np
import pytest
import torch

from cattus_amd import synth
from cattus_amd.torch_model import PolicyValueNet, SimpleTwoHeaded
from cattus_amd.weights import CHESS, TTT, NetDesc, blob_from_module, blob_from_state_dict, hex_game, seeded_blob, simple_desc, state_dict_from_blob
from oracle import oracle

from helpers import GOLDEN, outputs_equal_ref_tol

SIMPLE = sorted(p.stem for p in (GOLDEN / "simple").glob("*.npz"))


def _load(name):
    z = np.load(GOLDEN / "simple" / f"{name}.npz")
    d = NetDesc(*[int(x) for x in z["desc"]])
    return d, seeded_blob(d, int(z["seed"])), z


def test_fixtures_present():
    assert SIMPLE == ["chess_simple", "hex11_simple", "hex5_simple", "ttt_simple"]


@pytest.mark.parametrize("name", SIMPLE)
def test_oracle_matches_reference_simple_model(name):
    d, blob, z = _load(name)
    assert d.simple and d.features == d.planes * d.board * d.board
    policy, value = oracle.OracleNet(blob).forward(z["planes"])
    assert outputs_equal_ref_tol(policy, value, z["policy"], z["value"])
    err_oracle = np.abs(policy - z["policy_f64"]).max()
    err_ref = np.abs(z["policy"] - z["policy_f64"]).max()
    assert err_oracle <= 4 * err_ref + 1e-6


def test_simple_blob_round_trip_and_key_names():
    d = simple_desc(**hex_game(5))
    blob = seeded_blob(d, 3)
    d2, sd = state_dict_from_blob(blob)
    assert d2 == d
    # the reference module's parameter names (net_utils.py:101-110)
    assert list(sd) == [f"{m}.{p}" for m in ("_dense1", "_dense2", "_value_head", "_policy_head") for p in ("weight", "bias")]
    k = d.features
    assert sd["_dense1.weight"].shape == (k, k) and sd["_policy_head.weight"].shape == (d.moves, k) and sd["_value_head.bias"].shape == (1,)
    assert blob_from_state_dict(d, sd) == blob
    net = PolicyValueNet.from_blob(blob)
    assert isinstance(net, SimpleTwoHeaded)
    # the trainer's export_model hook recognises the module by its keys
    assert blob_from_module(net, (1, d.planes, d.board, d.board)) == blob


@pytest.mark.parametrize("name", SIMPLE)
def test_torch_module_matches_reference_fixture(name):
    d, blob, z = _load(name)
    net = PolicyValueNet.from_blob(blob)
    with torch.no_grad():
        p, v = net(torch.from_numpy(oracle.planes_to_tensor(z["planes"], d.board, len(z["planes"]))))
    np.testing.assert_allclose(p.numpy(), z["policy"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(v.numpy().ravel(), z["value"], rtol=1e-5, atol=1e-6)


def test_oracle_rejects_a_malformed_simple_header():
    d = simple_desc(**TTT)
    blob = bytearray(seeded_blob(d, 1))
    with pytest.raises(Exception):
        oracle.OracleNet(bytes(blob[:-4]))  # one float short


# ---- GPU ------------------------------------------------------------------------------------------------------------


def _words(planes):
    return planes.shape[2]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16x2", "bf16"])
@pytest.mark.parametrize("name", SIMPLE)
def test_hip_simple_model_bit_exact_vs_oracle(name, dtype):
    """Dense layers run in f32 whatever dtype the evaluator is configured with, in the oracle's accumulation order."""
    from cattus_amd.evaluator import HipEvaluator

    d, blob, z = _load(name)
    planes = z["planes"]
    want_p, want_v = oracle.OracleNet(blob).forward(planes)
    with HipEvaluator(blob, batch_size=len(planes) + 2, plane_words=_words(planes), dtype=dtype) as ev:
        got_p, got_v = ev.eval(planes)
        one_p, one_v = ev.eval(planes[:1])
    assert (got_p == want_p).all(), f"max |dp| = {np.abs(got_p - want_p).max()}"
    assert (got_v == want_v).all()
    assert (one_p == want_p[:1]).all() and (one_v == want_v[:1]).all()
    assert outputs_equal_ref_tol(got_p, got_v, z["policy"], z["value"])


@pytest.mark.gpu
def test_hip_simple_model_full_batches_and_threads():
    """Ragged and full batches (batch rows are independent), and the blocking per-leaf entry point from many threads."""
    import threading

    from cattus_amd.evaluator import HipEvaluator

    d = simple_desc(**CHESS)
    blob = seeded_blob(d, 9)
    planes = synth.random_chess_planes(300, 5)
    want_p, want_v = oracle.OracleNet(blob).forward(planes, threads=8)
    with HipEvaluator(blob, batch_size=128, plane_words=1, dtype="f16x2") as ev:
        for lo, hi in ((0, 128), (128, 135), (135, 300 - 37), (300 - 37, 300)):
            for a in range(lo, hi, 128):
                b = min(hi, a + 128)
                p, v = ev.eval(planes[a:b])
                assert (p == want_p[a:b]).all() and (v == want_v[a:b]).all()
        out = {}

        def leaf(i):
            out[i] = ev.wait(ev.submit(planes[i]))

        ts = [threading.Thread(target=leaf, args=(i,)) for i in range(24)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for i in range(24):
            assert (out[i][0] == want_p[i]).all() and out[i][1] == want_v[i]


@pytest.mark.gpu
def test_hip_simple_model_self_play_records_match_the_oracle_search():
    """Self-play on a SimpleTwoHeadedModel (the reference's TTT configuration): HIP records == oracle records."""
    from cattus_amd import selfplay as sp
    from cattus_amd.evaluator import HipEvaluator

    d = simple_desc(**TTT)
    blob = seeded_blob(d, 4)
    cfg = sp.make_config(sim_num=40, batch_size=4, threads=1, concurrent_games=1, cache_size=0, seed=5)
    onet = oracle.OracleNet(blob)
    with HipEvaluator(blob, batch_size=4, plane_words=1, dtype="f16x2") as ev:
        got = sp.run_self_play("tictactoe", cfg, sp.Net.hip(ev), None, 4, keep_records=True)
    want = sp.run_self_play("tictactoe", cfg, sp.Net.python(lambda pl: onet.forward(pl.reshape(len(pl), d.planes, 1), threads=1)), None, 4,
                            keep_records=True)
    assert got["records"] == want["records"] > 0
    assert (got["record_bytes"] == want["record_bytes"]).all() and (got["record_meta"] == want["record_meta"]).all()
