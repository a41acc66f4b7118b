"""twilight_amd -- MI355X-native level-batch aligner for TWILIGHT's TALCO-XDrop hot path.

The product is ``libtwl_align.so`` (hand-written HIP for gfx950 behind the C ABI of
``include/twl_align.h``).  This package is a thin ctypes binding used by the tests and by
``bench.py``; it has no CPU fallback: if the shared library is missing or no GPU is present the
calls raise.
"""
from .api import (  # noqa: F401
    LIB_PATH,
    TwlError,
    TwlParams,
    TwlStats,
    align_batch,
    align_batch_device,
    column_scores,
    dp_column_scores,
    exported_symbols,
    get_pair_cells,
    get_stats,
    init,
    load_library,
    make_params,
    set_knob,
    shutdown,
    version,
)
from . import api as knobs  # noqa: F401  (KNOB_* constants)

__version__ = "0.5.0"
