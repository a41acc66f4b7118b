"""One line per level of a verbose run (development): TWL_BENCH_VERBOSE=1 python bench.py ... 2> run.err; python tools/level_table.py run.err <levels>
Columns: level, pairs, wall ms of the level | host prepare ms (of which device kernels), level-kernel call ms (of which DP kernels), finish ms (device
write-back; verbose runs wait for it), whole | pairs that lost columns, pairs handed back to the host, host ms for gappy columns.  The last <levels>
levels of the log are shown (= the last pass)."""
import re,sys
f=sys.argv[1]; nl=int(sys.argv[2])
L=open(f).read().splitlines()
rows=[];cur=None
for l in L:
    m=re.match(r"\s+phases \(ms\): prepare ([\d.e+-]+) \(device ([\d.e+-]+)\) call ([\d.e+-]+) \(kernel ([\d.e+-]+), exchange ([\d.e+-]+)\) finish ([\d.e+-]+) \(device ([\d.e+-]+)\) whole ([\d.e+-]+); relaunched pairs (\d+); pairs with removed columns (\d+); restored on the host (\d+); gappy columns back ([\d.e+-]+)",l)
    if m: cur=[float(x) for x in m.groups()]
    m2=re.match(r"Level (\d+), aligned (\d+) pairs? in (\d+) ms",l)
    if m2 and cur:
        rows.append((int(m2.group(1)),int(m2.group(2)),int(m2.group(3)),cur)); cur=None
rows=rows[-nl:]
tot=[0]*6
print("lvl pairs lvlms | prep(dev) call(kern) fin(dev) whole | rm hb gappy")
for lv,n,ms,c in rows:
    print(f"{lv:3d} {n:6d} {ms:5d} | {c[0]:7.2f}({c[1]:6.2f}) {c[2]:7.2f}({c[3]:7.2f}) {c[5]:7.2f}({c[6]:6.2f}) {c[7]:7.2f} | {int(c[9])} {int(c[10])} {c[11]:.2f}")
    tot[0]+=c[0];tot[1]+=c[1];tot[2]+=c[2];tot[3]+=c[3];tot[4]+=c[5];tot[5]+=c[6]
print("tot prep %.1f (dev %.1f) call %.1f (kernel %.1f) finish %.1f (dev %.1f) sum whole %.1f sum level ms %d"%(*tot,sum(r[3][7] for r in rows),sum(r[2] for r in rows)))
