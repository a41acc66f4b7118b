"""Offline model of the two-lane schedule (development; reads a TWL_DUMP_SCHEDULE dump: level idx repA repB R Q err cells).

    python tools/sim_schedule.py dump.txt [cuL ...]

Lane L: cuL CUs, speculative kernel (2 CUs per pair), small batches of the most urgent ready pairs.
Lane T: the other CUs, every other ready pair.  Batches finish as a whole (host commit), which is what the model is for.
"""
import sys, heapq
import numpy as np

rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
N = len(rows)
last = {}
dep = [[] for _ in range(N)]
for k, (lv, i, a, b, R, Q, err, cells) in enumerate(rows):
    for s in (a, b):
        if s in last: dep[k].append(last[s])
    last[a] = k; last[b] = k
R = np.array([r[4] for r in rows]); Q = np.array([r[5] for r in rows]); cells = np.array([r[7] for r in rows], dtype=float)
lvl = np.array([r[0] for r in rows])
users = [[] for _ in range(N)]
for k in range(N):
    for d in dep[k]: users[d].append(k)
tS = (R + Q) * 0.65e-6; tL = (R + Q) * 0.93e-6
height = np.zeros(N)
for k in range(N - 1, -1, -1):
    height[k] = tS[k] + max([height[u] for u in users[k]], default=0.0)

def batch_time(idx, cus, ovh_fixed=2e-3, ovh_pair=5e-6):
    n = len(idx)
    if n == 0: return 0.0
    o = ovh_fixed + ovh_pair * n
    if 2 * n <= cus: return o + tS[idx].max()
    if n <= cus: return o + tL[idx].max()
    slots = 2 * cus
    rate = 150e9 / 512          # cells/s of one of 512 concurrently running workgroups
    t = np.sort(cells[idx] / rate)[::-1]
    h = [0.0] * slots
    heapq.heapify(h)
    for x in t: heapq.heappush(h, heapq.heappop(h) + x)
    return o + max(h)

def levelwise():
    tot = 0
    for l in range(lvl.max() + 1): tot += batch_time(np.where(lvl == l)[0], 256)
    return tot

def simulate(cuL, capT=100000, verbose=False):
    ndep = np.array([len(d) for d in dep])
    ready = set(np.where(ndep == 0)[0])
    done = 0; t = 0.0
    busy = {}        # lane -> (finish time, idx)
    nb = 0
    while done < N:
        for lane in ("L", "T"):
            if lane in busy or not ready: continue
            r = sorted(ready, key=lambda k: -height[k])
            other_busy = ("T" if lane == "L" else "L") in busy
            if lane == "L":
                if cuL == 0: continue
                take = r[:cuL // 2]
                cus = cuL if (other_busy or len(r) > len(take)) else 256
            else:
                # T leaves the most urgent ones to L when L is free to take them
                skip = 0 if ("L" in busy or cuL == 0) else min(len(r), cuL // 2)
                take = r[skip:skip + capT]
                if not take: continue
                cus = 256 - cuL if (other_busy or skip > 0 or cuL == 0) else 256
                if cuL == 0: cus = 256
            take = np.array(take)
            for k in take: ready.discard(k)
            busy[lane] = (t + batch_time(take, cus), take)
            nb += 1
        lane = min(busy, key=lambda x: busy[x][0])
        t, take = busy.pop(lane)
        for k in take:
            done += 1
            for u in users[k]:
                ndep[u] -= 1
                if ndep[u] == 0: ready.add(u)
    return t, nb

print("pairs", N, "level-synchronous model: %.3f s" % levelwise())
for cuL in [int(x) for x in sys.argv[2:]] or [0, 16, 32, 64, 96, 128]:
    for capT in (100000, 2048, 1024):
        t, nb = simulate(cuL, capT)
        print("cuL %3d capT %6d: %.3f s in %d batches" % (cuL, capT, t, nb))

def dataflow(ovh=0.3e-3):
    """Per-pair dataflow bound: a pair starts when its children are done and half a CU is free; it takes a whole CU pair (speculative),
    one CU or half a CU depending on how much is waiting."""
    ndep = np.array([len(d) for d in dep])
    ready = [(-height[k], k) for k in np.where(ndep == 0)[0]]
    heapq.heapify(ready)
    free = 512          # half-CUs
    running = []        # (finish, k, halfcus)
    t = 0.0; done = 0
    while done < N:
        while ready and free > 0:
            backlog = len(ready)
            if backlog * 4 <= free and free >= 4: need, dur = 4, tS
            elif backlog * 2 <= free and free >= 2: need, dur = 2, tL
            else: need, dur = 1, None
            _, k = heapq.heappop(ready)
            d = (cells[k] / (150e9 / 512)) if dur is None else dur[k]
            free -= need
            heapq.heappush(running, (t + d + ovh, k, need))
        t, k, need = heapq.heappop(running)
        free += need; done += 1
        for u in users[k]:
            ndep[u] -= 1
            if ndep[u] == 0: heapq.heappush(ready, (-height[u], u))
    return t

print("per-pair dataflow bound: %.3f s" % dataflow())
