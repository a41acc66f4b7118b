"""The end-to-end variants that the independent replay (oracle/msa_replay.py), the CPU checker (oracle/e2e_oracle) and the product CLI must
agree on: (name, family generator arguments, optional foreign insertion, CLI flags, environment).  Shared by
tests/golden/make_e2e_variants.py (writes the fixture from the REPLAY), tests/test_replay_cpu.py and tests/test_gpu_variants.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOW_TH = ["--test-cal-profile-th", "3", "--test-update-seq-th", "5"]      # small trees reach the cached-profile / compressed-group branches (development flags, include/twl_msa.h)

VARIANTS = [
    ("nuc_default", dict(leaves=40, length=400, P=6, seed=7, sub=0.08, indel=0.02), None, [], {}),
    ("nuc_r0.7", dict(leaves=40, length=400, P=6, seed=7, sub=0.08, indel=0.02), None, ["-r", "0.7"], {}),
    ("nuc_r1", dict(leaves=40, length=400, P=6, seed=7, sub=0.08, indel=0.02), None, ["-r", "1"], {}),
    ("nuc_wildcard", dict(leaves=40, length=400, P=6, seed=7, sub=0.08, indel=0.02), {"ambig": 0.05}, ["-w"], {}),
    ("nuc_ambiguous_letters_no_wildcard", dict(leaves=40, length=400, P=6, seed=7, sub=0.08, indel=0.02), {"ambig": 0.05}, [], {}),
    ("nuc_gap_penalties", dict(leaves=30, length=300, P=6, seed=11, sub=0.06, indel=0.02), None, ["--gap-open", "-30", "--gap-extend", "-3", "--match", "10"], {}),
    ("prot_default", dict(leaves=30, length=200, P=22, seed=9, sub=0.1, indel=0.02), None, [], {}),
    ("prot_blosum80_wildcard", dict(leaves=30, length=200, P=22, seed=9, sub=0.1, indel=0.02), {"ambig": 0.05}, ["-b", "80", "-w"], {}),
    ("prot_blosum45", dict(leaves=24, length=250, P=22, seed=19, sub=0.15, indel=0.02), None, ["-b", "45"], {}),
    ("nuc_length_deviation_deferrals", dict(leaves=50, length=300, P=6, seed=17, sub=0.08, indel=0.03), None, ["--length-deviation", "0.03"], {}),
    ("nuc_length_deviation_filter", dict(leaves=50, length=300, P=6, seed=17, sub=0.08, indel=0.03), None, ["--length-deviation", "0.03", "--filter"], {}),
    ("nuc_cache_and_compress", dict(leaves=50, length=300, P=6, seed=17, sub=0.08, indel=0.03), None, LOW_TH, {}),
    ("nuc_deferrals_cache_compress", dict(leaves=50, length=300, P=6, seed=17, sub=0.08, indel=0.03), None, ["--length-deviation", "0.05", "-r", "0.8", "--test-cal-profile-th", "2", "--test-update-seq-th", "3"], {}),
    ("prot_cache_and_compress", dict(leaves=30, length=200, P=22, seed=9, sub=0.1, indel=0.02), None, LOW_TH, {}),
    # families large enough to reach the cached-profile (>= 1000 sequences, msa.hpp:179) and compressed-group (> 1000 members, :180) branches
    # with the reference's own thresholds: the host mirror at those branches is checked by the independent replay, not only by itself
    ("nuc_2400_leaves_default_thresholds", dict(leaves=2400, length=600, P=6, seed=31, sub=0.03, indel=0.004), None, [], {}),
    ("prot_2200_leaves_default_thresholds", dict(leaves=2200, length=220, P=22, seed=33, sub=0.03, indel=0.004), None, [], {}),
    # more than 10 000 sequences on one side of the top pairs (the root splits 10 235 / 265): gapCharScore 0 for those pairs (alignment-cpu.cpp:88) and the
    # compressed-group bookkeeping (alignment-helper.cpp:479-500) at that size, checked by the independent replay and not only by the self-linked CPU checker
    ("nuc_10500_leaves_more_than_10000_on_one_side", dict(leaves=10500, length=150, P=6, seed=117, sub=0.03, indel=0.004), None, [], {}),
    ("nuc_xdrop_failure_retried_in_deferred_pass", dict(leaves=8, length=1200, P=6, seed=3, sub=0.03, indel=0.003), (2, 600, 4500), [], {}),
]


def write_family(d, fam, insertion):
    import numpy as np

    sys.path.insert(0, ROOT)
    from twilight_amd import synth

    nwk, seqs = synth.make_family(fam["leaves"], fam["length"], P=fam["P"], seed=fam["seed"], sub=fam["sub"], indel=fam["indel"])
    if isinstance(insertion, dict):        # a share of the residues becomes the ambiguity letter (N / X): what -w scores
        rng = np.random.default_rng(2)
        amb = "N" if fam["P"] == 6 else "X"
        seqs = [(name, "".join(amb if r < insertion["ambig"] else c for c, r in zip(s, rng.random(len(s))))) for name, s in seqs]
    elif insertion:
        k, at, n = insertion
        rng = np.random.default_rng(1)
        name, s = seqs[k]
        seqs[k] = (name, s[:at] + "".join("ACGT"[c] for c in rng.integers(0, 4, size=n)) + s[at:])
    t, f = os.path.join(d, "t.nwk"), os.path.join(d, "s.fa")
    open(t, "w").write(nwk + "\n")
    with open(f, "w") as fh:
        for name, s in seqs:
            fh.write(f">{name}\n{s}\n")
    return t, f, ("n" if fam["P"] == 6 else "p")
