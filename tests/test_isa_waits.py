"""The hand-placed `s_waitcnt vmcnt` of the forward kernels, checked in the EMITTED gfx950 code (CPU test, no GPU).

`dualnet_fwd_w1d_kernel` (and the 19x19 banded kernels) request weight fragments by inline asm into AGPRs and guard them
with explicit waits; an MFMA is not a memory operation, so only a `sched_barrier` keeps hipcc from hoisting it above the wait
(it happened once, and one self-play game in 44 came out different).  `tools/isa_check.py` replays the
disassembly of every kernel over its control-flow graph; this test runs it on the objects `tamago_amd.build` just made.
"""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_check  # noqa: E402


def _listing(lines):
    """instruction tuples as isa_check.kernels() yields them, from 'mnemonic operands [-> label]' lines"""
    labels, insts = {}, []
    addr = 0x100
    rows = []
    for ln in lines:
        if ln.endswith(":"):
            labels[ln[:-1]] = addr
            continue
        rows.append((addr, ln))
        addr += 8
    for a, ln in rows:
        tgt = None
        if "->" in ln:
            ln, lab = [t.strip() for t in ln.split("->")]
            tgt = labels[lab]
        parts = ln.split(None, 1)
        insts.append((a, parts[0], parts[1] if len(parts) > 1 else "", tgt))
    return insts


def test_checker_sees_an_mfma_hoisted_above_its_wait():
    good = _listing(["global_load_dwordx4 a[0:3], v0, s[0:1]",
                     "global_load_dwordx4 a[4:7], v0, s[0:1] offset:1024",
                     "s_waitcnt vmcnt(1)",
                     "v_mfma_f32_16x16x32_f16 v[8:11], a[0:3], v[0:3], 0",
                     "s_waitcnt vmcnt(0)",
                     "v_mfma_f32_16x16x32_f16 v[8:11], a[4:7], v[0:3], v[8:11]",
                     "s_endpgm"])
    assert isa_check.check_kernel(good)[0] == []
    # the bug of round 4: the second MFMA in front of the wait that guards its fragment
    bad = [good[0], good[1], good[2], good[3], good[5], good[4], good[6]]
    bad = [(0x100 + 8 * i,) + t[1:] for i, t in enumerate(bad)]
    viol, _ = isa_check.check_kernel(bad)
    assert len(viol) == 1 and viol[0][2] in ("a4", "a5", "a6", "a7") and "v_mfma" in viol[0][1]
    # a wait that counts too generously (vmcnt(2) with two loads behind the fragment's)
    lax = _listing(["global_load_dwordx4 a[0:3], v0, s[0:1]",
                    "global_load_dword v20, v1, s[2:3]",
                    "global_load_dword v21, v1, s[2:3] offset:4",
                    "s_waitcnt vmcnt(2)",
                    "v_mfma_f32_16x16x32_f16 v[8:11], a[0:3], v[0:3], 0",
                    "v_add_f32_e32 v22, v20, v21",
                    "s_endpgm"])
    viol, _ = isa_check.check_kernel(lax)
    assert [v[2] for v in viol] == ["v20"]                 # a[0:3] has returned (two newer loads), v20 / v21 have not


def test_checker_follows_the_loop_back_edge():
    # a fragment requested at the bottom of one iteration and used at the top of the next
    body = ["top:",
            "v_mfma_f32_16x16x32_f16 v[8:11], a[0:3], v[0:3], 0",
            "global_load_dwordx4 a[0:3], v0, s[0:1]",
            "s_cbranch_scc1 0 -> top",
            "s_endpgm"]
    viol, _ = isa_check.check_kernel(_listing(body))
    assert len(viol) == 1 and viol[0][0] == 0x100
    fixed = body[:3] + ["s_waitcnt vmcnt(0)"] + body[3:]
    assert isa_check.check_kernel(_listing(fixed))[0] == []


@pytest.fixture(scope="module")
def objects():
    from tamago_amd import build
    build.build(verbose=False)
    return build.OBJ_DIR


@pytest.mark.parametrize("obj, name, min_agpr_loads", [
    ("net_forward_w1d.hip.o", "dualnet_fwd_w1d_kernel", 100),
    ("net_forward_band.hip.o", "dualnet_fwd_band_kernel", 0),
    ("net_forward_w1dband.hip.o", "dualnet_fwd_w1dband_kernel", 100),
    ("net_forward_split.hip.o", "dualnet_fwd_split_kernel", 0),
])
def test_no_instruction_reads_a_register_whose_load_is_in_flight(objects, obj, name, min_agpr_loads):
    path = os.path.join(objects, obj)
    if not os.path.exists(path):
        pytest.skip(f"{obj} is not part of this build")
    found = 0
    for mangled, (viol, stats) in isa_check.check_object(path, name).items():
        # the PROF = true instantiations (s_memtime stamps for tools/phase_profile*.py) are timing-only builds: with the
        # stamps' extra registers hipcc moves AGPRs that are still in flight - their RESULTS are not used anywhere
        if "ELb1EE" in mangled or "ILb1EE" in mangled:
            continue
        found += 1
        assert stats["mfma"] > 0
        assert stats["agpr_loads"] >= min_agpr_loads, (mangled, stats)      # (the check is not vacuous)
        assert viol == [], f"{mangled}: " + "; ".join("0x%x %s reads %s" % v for v in viol)
    assert found > 0
