"""Randomized campaign on the THROUGHPUT geometries (levels of more pairs than CUs): the pairs of a random nucleotide case of tests/test_gpu_fuzz.py, replicated to
320 pairs, through the 768-row geometry (TWL_KNOB_THR_SMALL 1) and through the 512-row one (2: what outgrows it re-runs on 768 rows and on from there), each
against the oracle:  python tools/fuzz_thr.py START COUNT"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_lib as O
import twilight_amd as twl
from twilight_amd import api, synth
from test_gpu_fuzz import random_case

start, count = int(sys.argv[1]), int(sys.argv[2])
twl.init([0])
twl.set_knob(twl.knobs.KNOB_POISON_TB, 1)      # a traceback word whose store was wrongly skipped must read as garbage, not as zeros
bad = ran = small_reruns = 0
for seed in range(start, start + count):
    batch, matrix, pk = random_case(seed)
    if batch.P != 6:
        continue
    rng = np.random.default_rng(seed + 7)
    matrix = synth.nucleotide_matrix() if rng.random() < 0.7 else synth.nucleotide_matrix(match=10, mismatch=-9, transition=-9)      # the structured modes (2, 5) are what the throughput geometries serve
    if rng.random() < 0.3:
        pk["xdrop"] = int(rng.choice([7000, 9000, 12000]))       # bands that outgrow the windows
    idx = np.arange(320) % batch.n_pairs
    big = synth.LevelBatch(P=batch.P, seq_len=batch.seq_len, freq=batch.freq[idx], gap_open=batch.gap_open[idx], gap_extend=batch.gap_extend[idx], len=batch.len[idx], num=batch.num[idx])
    oa, on, oerr, ost = O.align_batch(O.make_params(matrix, **pk), batch, threads=4)
    onehot = int(all(int(x) == 1 for x in batch.num[:, 1]) and rng.random() < 0.5)
    for knob in (1, 2):
        twl.set_knob(api.KNOB_THR_SMALL, knob)
        twl.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, onehot)
        try:
            aln, n, err = twl.align_batch(twl.make_params(matrix, **pk), big)
            st = twl.get_stats(0)
        finally:
            twl.set_knob(api.KNOB_ASSUME_ONEHOT_QUERY, 0)
        ok = np.array_equal(err, oerr[idx]) and np.array_equal(n, on[idx]) and all(np.array_equal(aln[i, : n[i]], oa[idx[i], : on[idx[i]]]) for i in range(320))
        if knob == 2:
            small_reruns += int(st.n_relaunched > 0)
        if not ok:
            bad += 1
            print(f"FAIL seed {seed} knob {knob} onehot {onehot} kernel {bytes(st.kernel)[:40]} params {pk} len {batch.len.tolist()}", flush=True)
    ran += 1
twl.set_knob(api.KNOB_THR_SMALL, 0)
print(f"fuzz_thr seeds {start}..{start+count-1}: {ran} nucleotide cases x 2 geometries, {bad} failures; {small_reruns} cases with pairs that outgrew the 512-row window", flush=True)
sys.exit(1 if bad else 0)
