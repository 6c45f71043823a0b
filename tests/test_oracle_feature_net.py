"""Oracle featurise + DualNet forward vs reference-generated fixtures."""
import numpy as np
import pytest
import torch

from oracle.board import GoBoard
from oracle.feature import generate_input_planes
from oracle.net import OracleNet, make_state_dict, forward_logits
from tests.helpers import load_npz, oracle_replay


@pytest.mark.parametrize("size", [9, 13, 19])
def test_feature_planes(size):
    fix = load_npz(f"feat_s{size}.npz")
    brd = load_npz(f"board_s{size}.npz")
    specials = fix["special_seqs"]
    for i in range(len(fix["game"])):
        g, ply, color = int(fix["game"][i]), int(fix["ply"][i]), int(fix["color"][i])
        if g >= 0:
            board = oracle_replay(size, brd[f"g{g}_move"], brd[f"g{g}_color"], ply)
        else:
            seq = [int(v) for v in specials[-g - 1] if v != -9]
            board = GoBoard(size)
            c = 1
            for mv in seq:
                board.put_stone(mv, c)
                c = 3 - c
        planes = generate_input_planes(board, color)
        assert planes.dtype == np.float32 and planes.shape == (6, size, size)
        assert np.array_equal(planes, fix["planes"][i].astype(np.float32)), i


@pytest.mark.parametrize("size", [9, 13, 19])
def test_dualnet_forward(size):
    fix = load_npz(f"net_s{size}.npz")
    for seed in (0, 7):
        sd = make_state_dict(size, seed, float(fix[f"w{seed}_gain"]))
        x = torch.from_numpy(fix[f"w{seed}_planes"].astype(np.float32))
        net = OracleNet(sd)
        logits, vlogits = forward_logits(sd, x)
        # same ops, same library, but possibly another CPU: allow a few ulp of fp32
        assert np.allclose(logits.numpy(), fix[f"w{seed}_logits"], atol=2e-5, rtol=1e-5)
        assert np.allclose(vlogits.numpy(), fix[f"w{seed}_vlogits"], atol=2e-5, rtol=1e-5)
        pol, val = net.inference(x)
        assert np.abs(pol.numpy() - fix[f"w{seed}_policy"]).max() < 1e-6
        assert np.abs(val.numpy() - fix[f"w{seed}_value"]).max() < 1e-6
        lg, val2 = net.inference_with_policy_logits(x)
        assert torch.equal(lg, logits) and torch.equal(val2, val)
        # fp32 forward vs the reference's own fp64 forward: error budget for the HIP path
        assert np.abs(logits.numpy() - fix[f"w{seed}_logits64"]).max() < 1e-4
