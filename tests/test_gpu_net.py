"""HIP DualNet forward + featurise kernels vs the oracle / reference goldens (needs a GPU).

Tolerance (BASELINE.json north_star): policy / value within 1e-4 in fp32."""
import numpy as np
import pytest
import torch

from tests.helpers import load_npz

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _net(size, sd):
    from tamago_amd.nn.network.dual_net import DualNet
    net = DualNet(torch.device("cuda:0"), size)
    net.load_state_dict(sd)
    return net


@pytest.mark.parametrize("size", [9, 13, 19])
def test_forward_vs_reference_golden(size):
    from oracle.net import make_state_dict
    fix = load_npz(f"net_s{size}.npz")
    for seed in (0, 7):
        sd = make_state_dict(size, seed, float(fix[f"w{seed}_gain"]))
        net = _net(size, sd)
        x = torch.from_numpy(fix[f"w{seed}_planes"].astype(np.float32))
        pol, val = net.inference(x)
        assert np.abs(pol.numpy() - fix[f"w{seed}_policy"]).max() < TOL
        assert np.abs(val.numpy() - fix[f"w{seed}_value"]).max() < TOL
        lg, val2 = net.inference_with_policy_logits(x)
        assert torch.equal(val, val2)
        ref = fix[f"w{seed}_logits"]
        assert np.abs(lg.numpy() - ref).max() < TOL * max(1.0, np.abs(ref).max())
        # closeness to the fp64 forward of the reference: the HIP fp32 path should be as
        # good as the reference's own fp32 path
        err_hip = np.abs(lg.numpy() - fix[f"w{seed}_logits64"]).max()
        err_ref = np.abs(ref - fix[f"w{seed}_logits64"]).max()
        assert err_hip < 4 * err_ref + 1e-6


@pytest.mark.parametrize("size,batches", [(9, [1, 2, 7, 256, 257, 770, 1539, 1600]), (13, [1, 5, 300]), (19, [1, 3, 64])])
def test_forward_vs_oracle_random_planes(size, batches):
    from oracle.net import OracleNet, make_state_dict
    sd = make_state_dict(size, 3, 1.4)
    net = _net(size, sd)
    ora = OracleNet(sd)
    rs = np.random.RandomState(5)
    for b in batches:
        x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, size, size)).astype(np.float32))
        pol, val = net.inference(x)
        rp, rv = ora.inference(x)
        assert np.abs(pol.numpy() - rp.numpy()).max() < TOL, b
        assert np.abs(val.numpy() - rv.numpy()).max() < TOL, b
        # device-resident entry point gives the same bits as the host one
        pd, vd = net.forward_device(x.cuda())
        torch.cuda.synchronize()
        assert torch.equal(pd.cpu(), pol) and torch.equal(vd.cpu(), val)


def test_forward_empty_and_errors():
    from oracle.net import make_state_dict
    from tamago_amd.lib import TamagoHipError
    net = _net(9, make_state_dict(9, 0))
    pol, val = net.inference(torch.zeros((0, 6, 9, 9)))
    assert pol.shape == (0, 82) and val.shape == (0, 3)
    with pytest.raises(ValueError):
        net.inference(torch.zeros((1, 6, 19, 19)))
    bad = make_state_dict(9, 0)
    del bad["blocks.3.conv1.weight"]
    with pytest.raises(KeyError):
        net.load_state_dict(bad)
    with pytest.raises(TamagoHipError):
        from tamago_amd.nn.network.dual_net import DualNet
        DualNet(torch.device("cpu"), 9)


@pytest.mark.parametrize("size", [9, 13, 19])
def test_featurize_kernel_vs_reference_golden(size):
    """tg_featurize_dev against planes recorded from nn/feature.py in the reference."""
    from oracle.board import GoBoard
    from tamago_amd import lib as tl
    from tests.helpers import oracle_replay
    lib = tl.load()
    fix = load_npz(f"feat_s{size}.npz")
    brd = load_npz(f"board_s{size}.npz")
    specials = fix["special_seqs"]
    cells, to_move, prev, moves, want = [], [], [], [], []
    for i in range(len(fix["game"])):
        g, ply, color = int(fix["game"][i]), int(fix["ply"][i]), int(fix["color"][i])
        if g >= 0:
            board = oracle_replay(size, brd[f"g{g}_move"], brd[f"g{g}_color"], ply)
        else:
            board = GoBoard(size)
            c = 1
            for mv in [int(v) for v in specials[-g - 1] if v != -9]:
                board.put_stone(mv, c)
                c = 3 - c
        cells.append(board.get_board_data())
        to_move.append(color)
        prev.append(board.record_pos(board.moves - 1))
        moves.append(board.moves)
        want.append(fix["planes"][i].astype(np.float32))
    n = len(cells)
    d_cells = torch.tensor(np.array(cells, dtype=np.uint8)).cuda()
    d_tm = torch.tensor(np.array(to_move, dtype=np.int8)).cuda()
    d_prev = torch.tensor(np.array(prev, dtype=np.int32)).cuda()
    d_moves = torch.tensor(np.array(moves, dtype=np.int32)).cuda()
    out = torch.full((n, 6, size, size), 7.0, dtype=torch.float32, device="cuda")
    tl.check(lib.tg_featurize_dev(size, d_cells.data_ptr(), d_tm.data_ptr(), d_prev.data_ptr(),
                                  d_moves.data_ptr(), n, out.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream), "tg_featurize_dev")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.array(want))


@pytest.mark.parametrize("size,algo", [(9, "direct"), (9, "wino"), (9, "split16"), (9, "w1d"),
                                       (19, "direct"), (19, "wino"), (19, "split16"), (19, "w1dband")])
def test_every_tower_algorithm_matches_the_oracle(algo, size, monkeypatch):
    """Implementations of the residual tower: exact-fp32 Winograd F(2x2,3x3) kernel (TG_FWD_ALGO=wino; the
    layer outputs of a 19x19 board passing through a global scratch image), exact-fp32 direct implicit GEMM
    (direct), and the split-operand kernels on the 16-bit matrix pipe (split16 = f16 x 2 pieces, one wave per SIMD,
    at both sizes; 19x19: one board per workgroup, residual image in an L2-resident scratch; w1d = Winograd F(2,3) along
    x only on the same operand pieces, the 9x9 default; w1dband = the same tower at 19x19, a board over two workgroups, the
    19x19 default since round 5).  All must agree with the oracle at every workgroup shape.
    (Round 5: the two-waves-per-SIMD and 2-D Winograd kernels, measured slower, left the library; git history, commit 04640d1.)"""
    from oracle.net import OracleNet, make_state_dict
    monkeypatch.setenv("TG_FWD_ALGO", algo)
    sd = make_state_dict(size, 7, 1.5)
    net = _net(size, sd)
    ora = OracleNet(sd)
    rs = np.random.RandomState(11)
    # 9x9: G = 1 / 2 / 3 boards per workgroup, ragged tails; 19x19: fewer / more boards than CUs
    for b in ((1, 5, 256, 300, 512, 1301) if size == 9 else (1, 3, 300)):
        x = torch.from_numpy(rs.randint(-1, 2, size=(b, 6, size, size)).astype(np.float32))
        rp, rv = ora.inference(x)
        pol, val = net.inference(x)
        assert np.abs(pol.numpy() - rp.numpy()).max() < TOL, (algo, b)
        assert np.abs(val.numpy() - rv.numpy()).max() < TOL, (algo, b)


@pytest.mark.parametrize("algo", ["w1d", "split16"])
def test_results_do_not_depend_on_the_launch_size(algo, monkeypatch):
    """A position's policy and value must come out the same bits whether it is evaluated alone, in a mini-batch of one-board
    workgroups or deep inside a launch of three-board workgroups: self-play games would otherwise depend on how the boards
    of a lock-step move are grouped (tests/test_gpu_fastpath.py::test_selfplay_move_schemes_play_the_same_games).  The one-
    and three-board variants of each kernel family do the same arithmetic in the same order."""
    from oracle.net import make_state_dict
    monkeypatch.setenv("TG_FWD_ALGO", algo)
    net = _net(9, make_state_dict(9, 5, 1.5))
    x = torch.from_numpy(np.random.RandomState(3).randint(-1, 2, size=(1000, 6, 9, 9)).astype(np.float32))
    big = net.inference_with_policy_logits(x)
    for lo, hi in ((0, 100), (7, 8), (500, 756), (997, 1000)):
        part = net.inference_with_policy_logits(x[lo:hi])
        assert torch.equal(part[0], big[0][lo:hi]) and torch.equal(part[1], big[1][lo:hi]), (algo, lo, hi)


def test_results_do_not_depend_on_what_another_stream_is_doing():
    """Two streams forward their own ragged batches (three-board groups + a tail of one-board workgroups) at the same time,
    over and over: every result equals the single-stream one bit for bit.  (Found with this: hipcc had moved the first
    MFMA that reads a layer's tap-2 fragments in front of the s_waitcnt that guards them - an MFMA is no memory operation,
    only a sched_barrier keeps it behind an inline-asm wait - and under another stream's memory traffic the fragments
    were late: tests/test_gpu_fastpath.py::test_selfplay_move_schemes_play_the_same_games turned flaky.)"""
    from oracle.net import make_state_dict
    net = _net(9, make_state_dict(9, 23, 1.5))
    x = torch.from_numpy(np.random.RandomState(11).randint(-1, 2, size=(1200, 6, 9, 9)).astype(np.float32)).cuda()
    xa, xb = x[:864], x[300:300 + 808].contiguous()
    ra, rb = net.forward_device(xa), net.forward_device(xb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(40):
        with torch.cuda.stream(s1):
            outs_a.append(net.forward_device(xa))
        with torch.cuda.stream(s2):
            outs_b.append(net.forward_device(xb))
            outs_b.append(net.forward_device(xb[:40]))
    torch.cuda.synchronize()
    assert all(torch.equal(p, ra[0]) and torch.equal(v, ra[1]) for p, v in outs_a)
    assert all(torch.equal(p, rb[0][:p.shape[0]]) and torch.equal(v, rb[1][:p.shape[0]]) for p, v in outs_b)


def test_banded_19x19_kernel_equals_the_one_workgroup_kernel(monkeypatch):
    """Small 19x19 launches spread a board over 4 (up to 64 boards) or 2 (up to 128) workgroups with halo rows exchanged
    through L2 (csrc/net_forward_band.hip): same arithmetic per output in the same order, so the same bits as the
    one-workgroup kernel that larger launches (and TG_FWD_BANDS=0) take - and the oracle's values within the tolerance."""
    from oracle.net import OracleNet, make_state_dict
    from tamago_amd import lib as tl
    lib = tl.load()
    sd = make_state_dict(19, 3, 1.4)
    net = _net(19, sd)
    x = torch.from_numpy(np.random.RandomState(8).randint(-1, 2, size=(140, 6, 19, 19)).astype(np.float32))
    monkeypatch.setenv("TG_FWD_ALGO", "split16")                   # (the default is the one-axis Winograd pair kernel, below)
    assert lib.tg_net_kernel_name(net.handle, 64).decode() == "dualnet_fwd_band_kernel<4>"
    assert lib.tg_net_kernel_name(net.handle, 100).decode() == "dualnet_fwd_band_kernel<2>"
    assert lib.tg_net_kernel_name(net.handle, 140).decode() == "dualnet_fwd_split_kernel<19, 1, f16x2>"
    whole = net.inference_with_policy_logits(x)                      # 140 boards: one workgroup per board
    four = net.inference_with_policy_logits(x[:64])
    two = net.inference_with_policy_logits(x[30:130])
    one = net.inference_with_policy_logits(x[139:140])
    assert torch.equal(four[0], whole[0][:64]) and torch.equal(four[1], whole[1][:64])
    assert torch.equal(two[0], whole[0][30:130]) and torch.equal(two[1], whole[1][30:130])
    assert torch.equal(one[0], whole[0][139:140]) and torch.equal(one[1], whole[1][139:140])
    for _ in range(5):                                               # the same launch again: nothing left behind in the flags
        again = net.inference_with_policy_logits(x[:64])
        assert torch.equal(again[0], four[0]) and torch.equal(again[1], four[1])
    monkeypatch.setenv("TG_FWD_BANDS", "0")
    assert lib.tg_net_kernel_name(net.handle, 64).decode() == "dualnet_fwd_split_kernel<19, 1, f16x2>"
    plain = net.inference_with_policy_logits(x[:64])
    assert torch.equal(plain[0], four[0]) and torch.equal(plain[1], four[1])
    monkeypatch.delenv("TG_FWD_BANDS")
    monkeypatch.setenv("TG_FWD_ALGO", "split16")
    rp, rv = OracleNet(sd).inference(x[:8])
    pol, val = net.inference(x[:8])
    assert np.abs(pol.numpy() - rp.numpy()).max() < TOL and np.abs(val.numpy() - rv.numpy()).max() < TOL
    # no launch above was redone by the exact kernel (a band that waits too long for its neighbour raises the range flag; the
    # one-workgroup kernel used to raise it on a stream's first launch: rows beyond the board read an unwritten scratch row)
    assert net.range_fallbacks() == 0


def test_banded_kernel_cannot_hang_a_silent_band_ends_in_the_exact_fallback(monkeypatch):
    """A band whose neighbour never shows up (here: band 1 keeps its sequence numbers to itself, TG_BAND_TEST_MUTE) gives up
    after the bounded wait, raises the range flag, and the exact-fp32 kernel queued behind the launch redoes the batch: the
    call returns within seconds, the result is the exact kernel's (oracle tolerance), the fallback is counted."""
    import time
    from oracle.net import OracleNet, make_state_dict
    sd = make_state_dict(19, 4, 1.2)
    net = _net(19, sd)
    x = torch.from_numpy(np.random.RandomState(9).randint(-1, 2, size=(4, 6, 19, 19)).astype(np.float32))
    monkeypatch.setenv("TG_FWD_ALGO", "split16")
    good = net.inference(x)
    assert net.range_fallbacks() == 0
    monkeypatch.setenv("TG_BAND_TEST_MUTE", "1")
    t0 = time.perf_counter()
    pol, val = net.inference(x)
    dt = time.perf_counter() - t0
    monkeypatch.delenv("TG_BAND_TEST_MUTE")
    assert dt < 20.0, dt
    assert net.range_fallbacks() == 1
    assert net.band_timeouts() >= 1                                  # ... told apart from an f16-range fallback
    rp, rv = OracleNet(sd).inference(x)
    assert np.abs(pol.numpy() - rp.numpy()).max() < TOL and np.abs(val.numpy() - rv.numpy()).max() < TOL
    assert np.abs(pol.numpy() - good[0].numpy()).max() < TOL
    # the next launch is an ordinary one - and, the device having shown that it cannot keep a board's bands resident, on the
    # one-workgroup kernel from now on (same bits)
    from tamago_amd import lib as tl
    assert b"band" not in tl.load().tg_net_kernel_name(net.handle, 4)
    again = net.inference(x)
    assert torch.equal(again[0], good[0]) and torch.equal(again[1], good[1]) and net.range_fallbacks() == 1


def test_a_shared_device_keeps_19x19_launches_on_the_one_workgroup_kernel(monkeypatch):
    """More shard processes than GPUs (selfplay/main.py, TG_SINGLE_DEVICE): another process's kernels can keep a board's bands
    off the CUs, so the banded kernel is not chosen (tg_net_set_shared_device) - same results, no 0.1 s bounded waits."""
    from oracle.net import make_state_dict
    from tamago_amd import lib as tl
    monkeypatch.delenv("TG_FWD_BANDS", raising=False)
    x = torch.from_numpy(np.random.RandomState(10).randint(-1, 2, size=(16, 6, 19, 19)).astype(np.float32))
    lib = tl.load()
    for algo in ("split16", None):                                   # the banded direct kernel by name, the pair kernel by default
        if algo:
            monkeypatch.setenv("TG_FWD_ALGO", algo)
        else:
            monkeypatch.delenv("TG_FWD_ALGO")
        net = _net(19, make_state_dict(19, 4, 1.2))
        assert b"band" in lib.tg_net_kernel_name(net.handle, 16)
        banded = net.inference(x)
        net.set_shared_device(True)
        assert lib.tg_net_kernel_name(net.handle, 16).decode() == "dualnet_fwd_split_kernel<19, 1, f16x2>"
        alone = net.inference(x)
        if algo:                                                     # (same arithmetic in the same order: same bits)
            assert torch.equal(alone[0], banded[0]) and torch.equal(alone[1], banded[1])
        else:                                                        # (Winograd vs direct: the tolerance)
            assert (alone[0] - banded[0]).abs().max() < TOL and (alone[1] - banded[1]).abs().max() < TOL
        assert net.band_timeouts() == 0 and net.range_fallbacks() == 0


def test_pair_kernel_19x19_launch_sizes_repeats_and_the_bounded_wait(monkeypatch):
    """dualnet_fwd_w1dband_kernel (the 19x19 default): a position's bits do not depend on the launch it is part of (1 board, fewer
    boards than pairs of CUs, several boards per pair), repeats leave nothing behind in the sequence numbers, and a partner
    band that never publishes (TG_WB_TEST_MUTE) ends in the bounded wait -> exact fallback, counted as a band time-out, after
    which the network stays on the one-workgroup kernel."""
    import time
    from oracle.net import OracleNet, make_state_dict
    from tamago_amd import lib as tl
    lib = tl.load()
    monkeypatch.delenv("TG_FWD_ALGO", raising=False)
    monkeypatch.delenv("TG_FWD_BANDS", raising=False)
    sd = make_state_dict(19, 6, 1.3)
    net = _net(19, sd)
    assert lib.tg_net_kernel_name(net.handle, 64).decode() == "dualnet_fwd_w1dband_kernel + dualnet_heads19_kernel"
    x = torch.from_numpy(np.random.RandomState(12).randint(-1, 2, size=(600, 6, 19, 19)).astype(np.float32))
    big = net.inference_with_policy_logits(x)
    for lo, hi in ((0, 40), (77, 78), (300, 500), (597, 600)):
        for _ in range(2):
            part = net.inference_with_policy_logits(x[lo:hi])
            assert torch.equal(part[0], big[0][lo:hi]) and torch.equal(part[1], big[1][lo:hi]), (lo, hi)
    rp, rv = OracleNet(sd).inference(x[:6])
    pol, val = net.inference(x[:6])
    assert np.abs(pol.numpy() - rp.numpy()).max() < TOL and np.abs(val.numpy() - rv.numpy()).max() < TOL
    assert net.range_fallbacks() == 0 and net.band_timeouts() == 0
    monkeypatch.setenv("TG_WB_TEST_MUTE", "1")
    t0 = time.perf_counter()
    pol2, val2 = net.inference(x[:6])
    dt = time.perf_counter() - t0
    monkeypatch.delenv("TG_WB_TEST_MUTE")
    assert dt < 20.0, dt
    assert net.range_fallbacks() == 1 and net.band_timeouts() >= 1
    assert np.abs(pol2.numpy() - rp.numpy()).max() < TOL and np.abs(val2.numpy() - rv.numpy()).max() < TOL
    assert b"band" not in lib.tg_net_kernel_name(net.handle, 6)
    net.inference(x[:6])
    assert net.range_fallbacks() == 1


def test_kernel_name_and_executed_flops_know_the_ragged_tail_split(monkeypatch):
    """A 9x9 launch whose remainder beyond whole rounds of three-board workgroups is at most one workgroup per CU goes out as TWO
    launches (three-board head + one-board tail): the name and the issued-FLOP figure bench.py prices the matrix pipe with
    must say so."""
    import ctypes
    from oracle.net import make_state_dict
    from tamago_amd import lib as tl
    monkeypatch.delenv("TG_FWD_ALGO", raising=False)
    net = _net(9, make_state_dict(9, 5, 1.5))
    lib = tl.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def flops(b):
        return lib.tg_net_executed_flops_per_position(net.handle, b, None, None)
    whole, ragged, small = 6 * cus, 6 * cus + 64, 64
    assert b"ragged tail" not in lib.tg_net_kernel_name(net.handle, whole)
    name = lib.tg_net_kernel_name(net.handle, ragged)
    assert b"ragged tail" in name and b"<3>" in name and b"<1>" in name
    want = (flops(whole) * whole + flops(small) * small) / ragged
    assert abs(flops(ragged) - want) < 1e-6 * want and flops(small) > flops(whole)


def test_split_kernels_are_fp32_class_and_fall_back_on_f16_overflow(monkeypatch):
    """Accuracy of every 9x9 kernel against the reference's own fp64 forward (tests/golden/net_s9.npz):
    the split-operand kernels must be as close to fp64 as the reference's fp32 CPU path is (same
    criterion as for the exact-fp32 kernels).  A network whose activations leave the f16 range makes
    the f16 kernel raise its range flag; the batch is then redone by the exact-fp32 kernel on the
    device - the result equals the Winograd kernel's bit for bit and is finite."""
    from oracle.net import OracleNet, make_state_dict
    fix = load_npz("net_s9.npz")
    errs = {}
    for algo in ("wino", "direct", "split16", "w1d"):
        monkeypatch.setenv("TG_FWD_ALGO", algo)
        worst = 0.0
        for seed in (0, 7):
            sd = make_state_dict(9, seed, float(fix[f"w{seed}_gain"]))
            net = _net(9, sd)
            x = torch.from_numpy(fix[f"w{seed}_planes"].astype(np.float32))
            n = x.shape[0]
            reps = (300 + n - 1) // n                  # above the CU count: the three-boards-per-workgroup kernels
            lg, _ = net.inference_with_policy_logits(x.repeat(reps, 1, 1, 1))
            lg = lg[:n]
            err_hip = np.abs(lg.numpy() - fix[f"w{seed}_logits64"]).max()
            err_ref = np.abs(fix[f"w{seed}_logits"] - fix[f"w{seed}_logits64"]).max()
            assert err_hip < 4 * err_ref + 1e-6, (algo, seed, err_hip, err_ref)
            worst = max(worst, err_hip / max(err_ref, 1e-12))
        errs[algo] = worst
    print("max |logit - fp64| relative to the reference fp32 path's:", errs)
    # f16 range guard
    sd = make_state_dict(9, 3, 1.4)
    sd["bn_layer.weight"] = sd["bn_layer.weight"] * 3000.0          # stem output ~1e4, tower beyond 6e4
    x = torch.from_numpy(np.random.RandomState(2).randint(-1, 2, size=(300, 6, 9, 9)).astype(np.float32))
    monkeypatch.setenv("TG_FWD_ALGO", "wino")
    want = _net(9, sd).inference_with_policy_logits(x)
    for algo in ("split16", "w1d"):
        monkeypatch.setenv("TG_FWD_ALGO", algo)
        hot = _net(9, sd)
        assert hot.range_fallbacks() == 0
        got = hot.inference_with_policy_logits(x)
        assert torch.isfinite(got[0]).all() and torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), algo
        # ... and the host can see that it happened (tg_net_range_fallbacks): one redone launch here, none for a healthy net
        assert hot.range_fallbacks() == 1, algo
        healthy = _net(9, make_state_dict(9, 3, 1.4))
        healthy.inference_with_policy_logits(x)
        assert healthy.range_fallbacks() == 0, algo


def test_one_hot_position_costs_one_groups_redo(monkeypatch):
    """The f16 range guard at group granularity (round 5): the one-axis Winograd kernels mark the workgroup passes whose activations
    left the f16 range, and the exact-fp32 kernel queued behind the launch redoes THOSE - three positions (a 9x9 group) or one
    board (19x19), not the launch.  Every other position keeps the bits of an undisturbed launch; the hot position gets the
    exact kernel's result; the counters say so (tg_net_range_fallbacks / tg_net_range_fallback_positions)."""
    from oracle.net import make_state_dict
    monkeypatch.delenv("TG_FWD_ALGO", raising=False)
    for size, n, hot in ((9, 1000, 77), (9, 200, 140), (19, 40, 5)):      # three-board groups; one-board workgroups; a 19x19 board
        sd = make_state_dict(size, 9, 1.4)
        x = torch.from_numpy(np.random.RandomState(21).randint(-1, 2, size=(n, 6, size, size)).astype(np.float32))
        net = _net(size, sd)
        clean = net.inference_with_policy_logits(x)
        assert net.range_fallbacks() == 0 and net.range_fallback_positions() == 0
        xh = x.clone()
        xh[hot] *= 3.0e4                                                  # stem output ~1e5: beyond the guard's 16 000
        got = net.inference_with_policy_logits(xh)
        group = 3 if (size == 9 and n > 256) else 1
        lo = hot - hot % group
        assert net.range_fallbacks() == 1 and net.range_fallback_positions() == group, (size, n)
        keep = torch.ones(n, dtype=torch.bool)
        keep[lo:lo + group] = False
        assert torch.equal(got[0][keep], clean[0][keep]) and torch.equal(got[1][keep], clean[1][keep])
        monkeypatch.setenv("TG_FWD_ALGO", "wino")
        exact = _net(size, sd).inference_with_policy_logits(xh[lo:lo + group])
        monkeypatch.delenv("TG_FWD_ALGO")
        assert torch.isfinite(got[0]).all()
        assert torch.equal(got[0][lo:lo + group], exact[0]) and torch.equal(got[1][lo:lo + group], exact[1])
        # the bitmap is clean again: the next launch redoes nothing
        again = net.inference_with_policy_logits(x)
        assert torch.equal(again[0], clean[0]) and net.range_fallbacks() == 1 and net.range_fallback_positions() == group
