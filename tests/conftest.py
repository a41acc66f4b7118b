import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the HIP library and the oracle once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def gpu(built):
    import twilight_amd as twl

    twl.init([0])
    # every DP launch of the GPU tests starts from a traceback scratch full of 0xFF bytes: a block stores its traceback word only when it can have held band cells,
    # and a word wrongly skipped must not look like the zeros of a fresh allocation (include/twl_align.h, TWL_KNOB_POISON_TB)
    twl.set_knob(twl.knobs.KNOB_POISON_TB, 1)
    yield twl
    twl.shutdown()
