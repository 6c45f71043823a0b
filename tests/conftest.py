import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# the library reads its kernel-variant / grouping / test-hook knobs (TG_SELECT_*, TG_SP_*, TG_FWD_BANDS, TG_*_TEST_MUTE ...) only
# in a process that asks for them (csrc/common.h tg::knob); the tests compare those variants, so they do
os.environ.setdefault("TG_DEBUG_KNOBS", "1")
# (what importing tamago_amd asks for; here as well so that it holds whatever is imported first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
