#!/usr/bin/env python3
"""Where a rank's host CPU time goes on configs[3]'s shard: the process is pinned to `share` CPUs, `inflight` lanes decode
the 8 192 single-signal segments `steps` times; CPU seconds by thread (/proc/self/task) and the library's own CPU
accounting by phase (wspr_last_timings [16..23]), per step.
usage: shard_cpu_profile.py [share] [inflight] [steps] [nseg] [nsig]"""
import ctypes as C
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

share = int(sys.argv[1]) if len(sys.argv) > 1 else 2
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
nseg = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
nsig = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if share > 0:
    os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:share])
    os.environ["WSPR_HOST_THREADS"] = str(share)
    os.environ["OMP_NUM_THREADS"] = str(share)
import torch  # noqa: E402
if share > 0:
    torch.set_num_threads(share)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import rtlsdr_wsprd_amd as w  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = w.lib()
if nsig == 1:
    I, Q, _ = bench.synth_batch_gpu(nseg, 1234, dev, 1, -20.0, -20.0, 1.0)
else:
    I, Q, _ = bench.synth_batch_gpu(nseg, 4321, dev, nsig, -10.0, -28.0, 0.3)
torch.cuda.synchronize()
lanes = [ThreadPoolExecutor(1) for _ in range(inflight)]


def bind(k):
    torch.cuda.set_device(0)
    L.wspr_set_thread_slots(1)
    return L.wspr_bind_thread_lane(k)


for k, ex in enumerate(lanes):
    ex.submit(bind, k).result()
decs = [w.BatchDecoder(nseg, 16 if nsig == 1 else 32) for _ in range(inflight)]
phase = {}


def one(k):
    decs[k].decode_ptr(I.data_ptr(), Q.data_ptr(), 45000, I.stride(0))
    return w.last_timings()


def run(n, collect=False):
    pend = []
    for s in range(n):
        if len(pend) >= inflight:
            t = pend.pop(0).result()
            if collect:
                for kk, v in t.items():
                    phase[kk] = phase.get(kk, 0.0) + v
        pend.append(lanes[s % inflight].submit(one, s % inflight))
    for f in pend:
        t = f.result()
        if collect:
            for kk, v in t.items():
                phase[kk] = phase.get(kk, 0.0) + v


def snap():
    d = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            d[t] = (comm, int(rest[11]), int(rest[12]))
        except Exception:
            pass
    return d


run(3 * inflight)
torch.cuda.synchronize()
a = snap()
t0 = time.time()
run(steps, True)
torch.cuda.synchronize()
wall = time.time() - t0
b = snap()
hz = os.sysconf("SC_CLK_TCK")
rows = []
for t, (comm, u, s) in b.items():
    u0, s0 = (a[t][1], a[t][2]) if t in a else (0, 0)
    rows.append(((u - u0 + s - s0) / hz, (u - u0) / hz, (s - s0) / hz, comm, t))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("share %d inflight %d nseg %d x %d signals: %.2f ms/step wall, %.0f segments/s; CPU %.1f ms/step (%.0f %% of %d CPUs)" % (
    share, inflight, nseg, nsig, wall / steps * 1e3, nseg * steps / wall, tot / steps * 1e3, 100 * tot / wall / max(1, share), share))
for r in rows[:10]:
    print("  %.2f s (user %.2f sys %.2f) %s tid %s" % r)
print("library accounting, ms per step:", json.dumps({k: round(v / steps, 2) for k, v in phase.items() if k.endswith("_ms") or k.startswith("cpu_ms")}))
