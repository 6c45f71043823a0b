"""CPU-only checks of the C ABI library and of the Python host layer (no GPU compute)."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from tamago_amd import build, lib as tl
    build.build(verbose=False)
    return tl.load()


def header_functions():
    text = open(os.path.join(REPO, "include", "tamago_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from tamago_amd import lib as tl
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tamago_hip.h but not exported"
    assert declared == tl.exported_symbols(), "ctypes signatures out of sync with the header"


def test_library_queries_without_gpu(lib):
    assert lib.tg_abi_version() >= 1
    assert lib.tg_net_param_count(9) == 462968       # 461 298 trainable + 1 670 BN buffers
    assert lib.tg_net_param_count(19) == 712168
    assert lib.tg_net_flops_per_position(9) == 2 * 36140823
    assert lib.tg_net_flops_per_position(19) == 2 * 161274223


def test_argument_errors_are_reported(lib):
    import ctypes
    from tamago_amd import lib as tl
    h = ctypes.c_void_p()
    rc = lib.tg_net_create(9, 0, None, 0, ctypes.byref(h))
    assert rc == -1 and b"null" in lib.tg_last_error()
    blob = np.zeros(10, dtype=np.float32)
    rc = lib.tg_net_create(11, 0, blob.ctypes.data, blob.size, ctypes.byref(h))       # (9, 13 and 19 are built)
    assert rc == -1 and b"board size 11" in lib.tg_last_error()
    rc = lib.tg_net_create(13, 0, blob.ctypes.data, blob.size, ctypes.byref(h))
    assert rc == -1 and b"parameters" in lib.tg_last_error()
    cfg = tl.SearchConfig(9, 0, 16, 1, 0, 0, 0, 0)
    rc = lib.tg_search_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == -1
    with pytest.raises(tl.TamagoHipError):
        tl.check(rc, "tg_search_create")


def test_state_dict_layout_matches_param_count(lib):
    from tamago_amd.nn.network.dual_net import state_dict_keys, random_state_dict
    from oracle.net import state_dict_shapes
    for size in (9, 13, 19):
        keys = state_dict_keys(size)
        assert sum(int(np.prod(s)) for _, s in keys) == lib.tg_net_param_count(size)
        ref = {k: v for k, v in state_dict_shapes(size).items() if not k.endswith("num_batches_tracked")}
        assert dict(keys) == ref
        sd = random_state_dict(size)
        assert set(sd) == set(ref)
