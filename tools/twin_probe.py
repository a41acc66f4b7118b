"""Experiment (round 6): TWO level schedules side by side on one MI355X in ONE process.  Two handles of the host library open the same family, each on a context of
its own of device 0 (ids 0 and 256: streams, scratch and lock of their own), sharded as ranks 0 and 1 of a world of 2 -- subtree ownership below the cut, one exchange at
the cut, the pairs of the levels above dealt -- with the all-gathers done in process (a barrier and copies).  Their align() calls run on two threads.

    python tools/twin_probe.py [leaves] [length] [passes]        -> ms per pass of one handle alone and of the twin, MSA md5 of both"""
import ctypes as C, hashlib, os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import twilight_amd as twl
from twilight_amd import msa, synth
from twilight_amd.dist import _DevBuf

leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sys.setrecursionlimit(1000000)
d = tempfile.mkdtemp(prefix="twl_twin_")
nwk, seqs = synth.make_family(leaves, length, P=6, seed=20260501, sub=0.015, indel=0.001)
tree, fasta = os.path.join(d, "t.nwk"), os.path.join(d, "s.fa")
open(tree, "w").write(nwk + "\n")
with open(fasta, "w") as f:
    for name, s in seqs: f.write(f">{name}\n{s}\n")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
twl.init([0, 256])


class Bus:
    def __init__(self):
        self.bar = threading.Barrier(2)
        self.slot = [None, None]

    def host(self, rank):
        def ex(send, n, recv):
            self.slot[rank] = send
            self.bar.wait()
            for r in range(2): C.memmove(recv + r * n, self.slot[r], n)
            self.bar.wait()
            return 0
        return ex

    def device(self, rank):
        def ex(send, n, recv):
            self.slot[rank] = send
            self.bar.wait()
            with torch.cuda.device(dev):
                out = torch.as_tensor(_DevBuf(recv, 2 * n), device=dev)
                for r in range(2): out[r * n:(r + 1) * n].copy_(torch.as_tensor(_DevBuf(self.slot[r], n), device=dev))
                torch.cuda.synchronize(dev)
            self.bar.wait()
            return 0
        return ex


def md5(p): return hashlib.md5(open(p, "rb").read()).hexdigest()

# one handle alone
single = []
for i in range(passes + 1):
    m = msa.Msa(["-t", tree, "-i", fasta, "-o", os.path.join(d, "one.aln"), "--gpu-index", "0"]); m.upload(); single.append(m)
single[0].align()
torch.cuda.synchronize()
t0 = time.perf_counter()
for m in single[1:]: m.align()
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / passes
single[-1].write(); one_md5 = md5(os.path.join(d, "one.aln"))
for m in single: m.close()

# the twin
twins = []
for i in range(passes + 1):
    bus = Bus()
    a = msa.Msa(["-t", tree, "-i", fasta, "-o", os.path.join(d, "twin0.aln"), "--gpu-index", "0"]); a.shard(0, 2, bus.host(0), exchange_device=bus.device(0)); a.upload()
    b = msa.Msa(["-t", tree, "-i", fasta, "-o", os.path.join(d, "twin1.aln"), "--gpu-index", "256"]); b.shard(1, 2, bus.host(1), exchange_device=bus.device(1)); b.upload()
    twins.append((a, b))


def run(pair):
    th = [threading.Thread(target=x.align) for x in pair]
    for t in th: t.start()
    for t in th: t.join()

run(twins[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for p in twins[1:]: run(p)
torch.cuda.synchronize()
t_twin = (time.perf_counter() - t0) / passes
twins[-1][0].write(); twins[-1][1].write()
print(f"{leaves} x {length}: one handle {t_one * 1e3:.1f} ms per pass (md5 {one_md5[:8]}); twin contexts {t_twin * 1e3:.1f} ms per pass (md5 {md5(os.path.join(d, 'twin0.aln'))[:8]} / {md5(os.path.join(d, 'twin1.aln'))[:8]})")
tot, lv = twins[-1][0].report()
print("rank 0 levels (pairs, kernel ms, level ms):", [(int(l.pairs), round(l.kernel_ms, 2), round(l.level_ms, 2)) for l in lv])
for a, b in twins: a.close(); b.close()
