"""Write a synthetic family (tree + FASTA) as bench.py does: python tools/make_family.py <dir> <config> [survey8d|calibrated]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
d, name = sys.argv[1], sys.argv[2]
cfg = dict(bench.CONFIGS[name]); cfg["workload"] = sys.argv[3] if len(sys.argv) > 3 else "calibrated"
cfg["index"] = {"rnasim1k_band512": 1, "rnasim10k": 2, "rnasim100k": 3, "protein5k": 4}[name]
os.makedirs(d, exist_ok=True)
print(*bench.write_family(cfg, d))
