"""The launch policy of the nucleotide path (twilight_amd/csrc/twl_align.hip: plan_nucleotide, a pure function of a call's facts) through
twl_plan_describe -- no GPU: which kernel family, matrix mode and window a level of n pairs gets on a 256-CU device."""
import numpy as np
import pytest

import twilight_amd as twl
from twilight_amd import api, synth

M = synth.nucleotide_matrix()


def _lens(n, length):
    return np.full((n, 2), length, dtype=np.int32)


@pytest.mark.parametrize("n,length,onehot,expect", [
    (1, 10000, False, "tile-parallel; mode 2; window 1024"),                       # a lone long pair: every tile at once
    (128, 10000, False, "tile-parallel; mode 2; window 1024"),                     # up to CUs / 2 pairs: always
    (200, 10000, False, "tile-parallel; mode 2; window 1024"),                     # a badly filled single round of the throughput kernel
    (900, 10000, False, "tile-parallel; mode 2; window 1024"),                     # 70 % of one round of 5 x 256 workgroups: tiles spread better
    (1200, 10000, False, "throughput; mode 2; window 512; bulk 1200 tail 0"),      # 94 % of one round
    (1301, 10000, False, "throughput; mode 2; window 512; bulk 1280 tail 21"),     # the remainder through the tile-parallel path
    (1986, 10000, False, "throughput; mode 2; window 512; bulk 1280 tail 706"),
    (3358, 10000, True, "throughput; mode 5; window 512; bulk 2560 tail 798"),     # a leaf level (one-letter query rows): it tries the small window
    (5000, 10000, False, "throughput; mode 2; window 512; bulk 5000 tail 0"),      # 91 % of the fourth round: stays
    (1301, 1600, False, "throughput; mode 2; window 512; bulk 1301 tail 0"),       # short pairs: five workgroups per CU on a 512-row window; 3-4 tiles: no tile-parallel remainder
    (33325, 1600, True, "throughput; mode 5; window 512 or 768 (a sample of the level decides)"),      # the leaf level of 100 000 x 1.6 kbp: 8+ rounds
    (4, 900, False, "speculative teams, 16 waves; mode 2"),                        # too few tiles to spread: two workgroups per pair
    (200, 900, False, "speculative teams, 8 waves x 2 blocks; mode 2"),
    (300, 900, False, "throughput; mode 2; window 512; bulk 300 tail 0"),
])
def test_plan_of_a_level(built, n, length, onehot, expect):
    got = api.plan_describe(twl.make_params(M), _lens(n, length), qry_onehot=onehot)
    assert got.startswith(expect), got


def test_plan_depends_on_the_matrix_and_on_the_streak(built):
    general = M.copy()
    general[0, 1] = -7.0                                    # no match / transition / transversion structure: mode 1 kernels
    assert "mode 1" in api.plan_describe(twl.make_params(general), _lens(2000, 5000))
    wild = synth.nucleotide_matrix()
    wild[4, :] = wild[:, 4] = 18.0                           # -w: N scores like a match -> the general 5 x 5 mode
    assert "mode 0" in api.plan_describe(twl.make_params(wild), _lens(2000, 5000))
    huge = M * 4096.0                                        # outside the hoisted-reciprocal division's range: the IEEE-division kernels
    assert api.plan_describe(twl.make_params(huge), _lens(10, 5000)).startswith("general (IEEE division)")
    # a streak of small calls whose pairs all outgrew the fast window starts on the wide one; every 8th call probes the fast window again
    assert api.plan_describe(twl.make_params(M), _lens(1, 12000), wide_streak=3).startswith("tile-parallel, 3072-row window; mode 2; window 3072")
    assert api.plan_describe(twl.make_params(M), _lens(1, 12000), wide_streak=7).startswith("tile-parallel; mode 2; window 1024")
    assert api.plan_describe(twl.make_params(M), _lens(20, 12000), wide_streak=3).startswith("tile-parallel; mode 2; window 1024")
    # ... and a level of up to CUs pairs starts wide when three quarters of the previous narrow-first level went on to the wide window; every 6th such call probes
    assert api.plan_describe(twl.make_params(M), _lens(100, 12000), wide_streak=1001).startswith("tile-parallel, 3072-row window")
    assert api.plan_describe(twl.make_params(M), _lens(100, 12000), wide_streak=1000).startswith("tile-parallel; mode 2; window 1024")
    assert api.plan_describe(twl.make_params(M), _lens(100, 12000), wide_streak=1051).startswith("tile-parallel; mode 2; window 1024")
    assert api.plan_describe(twl.make_params(M), _lens(300, 12000), wide_streak=1001).startswith("tile-parallel; mode 2; window 1024")      # (more pairs than CUs: the fast window first)


def test_a_level_that_outgrew_the_small_window_keeps_the_rest_of_the_pass_off_it(built):
    p = twl.make_params(M)
    assert "window 512;" in api.plan_describe(p, _lens(1500, 1600))                         # nothing remembered, a small level: it tries
    assert "window 768;" in api.plan_describe(p, _lens(1500, 1600), small_state=-1)
    assert "window 768;" in api.plan_describe(p, _lens(1500, 1600), small_state=-1)
    assert "window 512;" in api.plan_describe(p, _lens(1500, 1600), small_state=1)
    # a level of eight or more rounds with nothing remembered asks a sample of its own pairs
    assert "a sample of the level decides" in api.plan_describe(p, _lens(5000, 1600))
    assert "window 512;" in api.plan_describe(p, _lens(5000, 1600), small_state=1)
    assert "window 768;" in api.plan_describe(p, _lens(5000, 1600), small_state=-1)
    assert "a sample" not in api.plan_describe(p, _lens(5000, 10000))                    # long pairs are not sampled (a pair's latency): they try, or do as the level before them
    assert api.plan_describe(p, _lens(1986, 10000), small_state=-1).startswith("throughput; mode 2; window 768; bulk 1986 tail 0")
    assert api.plan_describe(p, _lens(3358, 10000), qry_onehot=True, small_state=-1).startswith("throughput; mode 5; window 768; bulk 3072 tail 286")
    twl.set_knob(api.KNOB_THR_SMALL, 1)
    try:
        assert "window 768;" in api.plan_describe(p, _lens(5000, 1600))
        twl.set_knob(api.KNOB_THR_SMALL, 2)
        assert "window 512;" in api.plan_describe(p, _lens(5000, 10000), small_state=-1)
    finally:
        twl.set_knob(api.KNOB_THR_SMALL, 0)


def test_every_knob_the_header_names_is_known_to_the_library_and_to_the_bindings(built):
    """include/twl_align.h's enum twl_knob, twilight_amd/api.py's KNOB_* constants and twl_set_knob's switch are three lists of the same keys."""
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "twl_align.h")).read()
    enum = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"TWL_(KNOB_[A-Z0-9_]+) = (\d+)", hdr[hdr.index("enum twl_knob"):]))
    assert len(enum) >= 20 and sorted(enum.values()) == list(range(1, len(enum) + 1))
    defaults = {"KNOB_MT_MAX_PAIRS": 1024, "KNOB_MT_MIN_MARKER": 512, "KNOB_MT_LEAD": 320, "KNOB_MT_MARGIN": -1, "KNOB_MT_ROUNDS": 2, "KNOB_MT_THR_JOBS": 256,
                "KNOB_MT_TAIL_PCT": 70, "KNOB_MT_WIDE": 1, "KNOB_SCOUT_XDROP_PCT": 100, "KNOB_LEAF_STEP": 1, "KNOB_MT_ANCHOR": 1, "KNOB_MT_LEAD2": -1, "KNOB_PROT_CORRIDOR": 448}
    for name, key in enum.items():
        assert getattr(api, name) == key, name
        twl.set_knob(key, defaults.get(name, 0))          # (every key is accepted; the value is the library's default)
    with pytest.raises(Exception):
        twl.set_knob(len(enum) + 1, 0)
