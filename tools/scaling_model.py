"""What the subtree-ownership design of DESIGN.md section 5 expects at N GPUs, from a single-GPU bench line (its per-level pairs and kernel times).
A MODEL, not a measurement (no multi-GPU node was available): python tools/scaling_model.py profiles/r04/bench_line_default.json [N ...]

Assumptions, all from the single-GPU level table:
  * the cut is the highest level that leaves >= 8 subtrees per rank (align_owned.cpp); subtrees are balanced by the longest-first deal;
  * below the cut a rank runs 1/N of every level's DP time, but never less per level than the level floor (one tile's latency + scouts: the smallest
    level time of the run), and only for levels in which an average subtree still has a pair (pairs_at_level >= subtrees);
  * the non-DP time of the pass divides by N below the cut and is replicated above it (in proportion to the pairs);
  * the exchange at the cut all-gathers the subtrees' rows HBM to HBM (rows ~1.5 x the sequence length at the cut; 100 GB/s effective per rank over xGMI)
    + 3 ms for the host blocks (node state, cached profiles) and the pack / unpack kernels; above the cut one all-gather per level (0.15 ms);
  * above the cut a level costs max(t / N, floor)."""
import json, sys

d = json.load(open(sys.argv[1]))
ranks = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
lv = d["levels"]
main = [l for l in lv if True]
pairs = [l["pairs"] for l in main]
kms = [l["kernel_ms"] for l in main]
floor = min(kms)
t1 = d["ms_per_step"]
nondp = t1 - sum(kms)
rows_bytes = d["config"]["n_sequences"] * d["config"]["seq_length"] * 1.5
print(f"single GPU: {t1:.1f} ms per pass, DP {sum(kms):.1f} ms over {len(kms)} levels, level floor {floor:.2f} ms, non-DP {nondp:.1f} ms")
for n in ranks:
    if n == 1:
        print(f"N = 1: {t1:.1f} ms (measured)")
        continue
    want = 8 * n
    above = 0
    cut = -1
    for l in range(len(pairs) - 2, -1, -1):
        if sum(pairs[l + 1:]) + 1 >= want:
            cut = l
            break
    if cut < 0:
        print(f"N = {n}: no cut (every level dealt per level)")
        continue
    subtrees = sum(pairs[cut + 1:]) + 1
    below = 0.0
    for l in range(cut + 1):
        if pairs[l] >= subtrees / 2:
            below += max(kms[l] / n, floor if pairs[l] < 4 * 256 * n else 0.0)
        else:
            below += floor * pairs[l] / max(1.0, subtrees / n) / 1.0 if pairs[l] * n < subtrees else floor
    top = sum(max(k / n, floor) for k in kms[cut + 1:])
    n_pairs = sum(pairs)
    nd_below = nondp * sum(pairs[:cut + 1]) / n_pairs / n
    nd_top = nondp * sum(pairs[cut + 1:]) / n_pairs
    exch = 3.0 + rows_bytes * (n - 1) / n / 100e9 * 1e3 + 0.15 * (len(kms) - cut - 1)
    t = below + top + nd_below + nd_top + exch
    print(f"N = {n}: cut after level {cut + 1} ({subtrees} subtrees, {sum(pairs[cut + 1:])} pairs above); below {below:.1f} + above {top:.1f} + non-DP {nd_below + nd_top:.1f} + exchange {exch:.1f} "
          f"= {t:.1f} ms -> {t1 / t:.2f}x")
