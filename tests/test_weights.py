"""Weight blob <-> state_dict converters and the torch cross-check module."""

import numpy as np
import pytest
import torch

from cattus_amd import weights as W
from cattus_amd.torch_model import PolicyValueNet
from oracle import oracle

from helpers import blob_for, golden_names


def test_blob_roundtrip_and_size():
    d = W.NetDesc(**W.CHESS, blocks=2, filters=8, vhc=4, phc=4)
    blob = W.seeded_blob(d, 5)
    assert len(blob) == W.blob_nbytes(d)
    d2, t = W.unpack_tensors(blob)
    assert d2 == d and W.pack_tensors(d, t) == blob
    assert W.seeded_blob(d, 5) == blob and W.seeded_blob(d, 6) != blob
    with pytest.raises(ValueError):
        W.unpack_tensors(blob[:-4])
    with pytest.raises(ValueError):
        W.parse_header(b"nope" + blob[4:])


def test_state_dict_keys_match_reference_checkpoint_layout():
    # key names of the reference's ConvNetV1 state_dict
    d = W.NetDesc(**W.hex_game(5), blocks=2, filters=8, vhc=4, phc=4)
    net = PolicyValueNet(d)
    keys = {k for k in net.state_dict() if not k.endswith("num_batches_tracked")}
    assert keys == {name for name, _ in W.tensor_specs(d)}
    blob = W.blob_from_state_dict(d, net.state_dict())
    net2 = PolicyValueNet.from_blob(blob)
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k])


@pytest.mark.parametrize("name", [n for n in golden_names() if n != "chess_20x256"])
def test_torch_module_agrees_with_reference_outputs_and_oracle(name):
    d, blob, z = blob_for(name)
    net = PolicyValueNet.from_blob(blob)
    x = torch.from_numpy(z["input_tensor"].astype(np.float32))
    with torch.no_grad():
        p, v = net(x)
    np.testing.assert_allclose(p.numpy(), z["policy"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v.numpy().ravel(), z["value"], rtol=1e-4, atol=1e-6)
    po, vo = oracle.OracleNet(blob).forward(z["planes"])
    np.testing.assert_allclose(po, p.numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(vo, v.numpy().ravel(), rtol=1e-4, atol=1e-5)


def test_flop_formula_matches_survey():
    assert W.NetDesc(**W.CHESS, blocks=20, filters=256, vhc=8, phc=8).flops_per_position() == 3027788032
    assert W.NetDesc(**W.hex_game(7), blocks=6, filters=64, vhc=16, phc=16).flops_per_position() == 43999904
    assert W.NetDesc(**W.CHESS, blocks=40, filters=384, vhc=8, phc=8).flops_per_position() == 13600350464


def test_blob_from_module_infers_the_shape():
    """export_model's hook for the hip engine: module -> blob, shape read off the state_dict."""
    d = W.NetDesc(planes=3, board=7, moves=49, blocks=3, filters=16, vhc=16, phc=16)
    net = PolicyValueNet.from_blob(W.seeded_blob(d, 3))
    assert W.desc_from_state_dict(net.state_dict(), 7) == d
    assert W.blob_from_module(net, (1, 3, 7, 7)) == W.seeded_blob(d, 3)
    with pytest.raises(ValueError):
        W.desc_from_state_dict(net.state_dict(), 5)
