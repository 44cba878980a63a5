#!/usr/bin/env python3
"""CPU seconds per thread over 300 decode calls of config 2 (the per-call slot driver threads have exited
by the time of the snapshot, so what is listed are the pool workers and the caller)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
import rtlsdr_wsprd_amd as w
dev=torch.device("cuda",0); torch.cuda.set_device(0)
I,Q,_=bench.synth_batch_gpu(1024, 1234, dev, 1, -20.0, -20.0, 1.0)
dec=w.BatchDecoder(1024, 16)
for _ in range(5): dec.decode(I,Q)
def snap():
    d={}
    for t in os.listdir("/proc/self/task"):
        try:
            f=open("/proc/self/task/%s/stat"%t).read()
            comm=f[f.index("(")+1:f.rindex(")")]; rest=f[f.rindex(")")+2:].split()
            d[t]=(comm,int(rest[11]),int(rest[12]))
        except Exception: pass
    return d
a=snap(); t0=time.time()
N=300
for _ in range(N): dec.decode(I,Q)
torch.cuda.synchronize(); wall=time.time()-t0
b=snap()
hz=os.sysconf("SC_CLK_TCK")
rows=[]
for t,(comm,u,s) in b.items():
    u0,s0=(a[t][1],a[t][2]) if t in a else (0,0)
    rows.append(((u-u0+s-s0)/hz, (u-u0)/hz, (s-s0)/hz, comm, t))
rows.sort(reverse=True)
print("wall %.2f s for %d steps (%.2f ms/step); CPU by thread:" % (wall, N, wall/N*1e3))
for r in rows[:14]: print("  %.2f s (user %.2f sys %.2f) %s tid %s" % r)
print("total CPU %.2f s = %.1f ms/step" % (sum(r[0] for r in rows), sum(r[0] for r in rows)/N*1e3))
