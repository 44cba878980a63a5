import ctypes as C, time, sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rtlsdr_wsprd_amd as w
from concurrent.futures import ThreadPoolExecutor
L = w.lib()
mt = (C.c_int * 256 * 2)(); L.wspr_fano_metric_table(mt)
rng = np.random.default_rng(1)
soft = rng.integers(60, 200, 162).astype(np.uint8)   # undecodable -> time-out
def one(_):
    s = (C.c_ubyte * 162)(*soft.tolist()); dec = (C.c_ubyte * 11)(); a = C.c_uint(); b = C.c_uint(); c = C.c_uint()
    r = L.fano(C.byref(a), C.byref(b), C.byref(c), dec, s, C.c_uint(81), mt, C.c_int(60), C.c_uint(10000))
    return r, b.value
t = time.perf_counter(); r = one(0); dt1 = time.perf_counter() - t
print("single timeout: ret", r, "ms", dt1 * 1e3)
for T in (1, 8, 32, 64, 128, 256):
    n = T * 8
    with ThreadPoolExecutor(T) as ex:
        t = time.perf_counter(); list(ex.map(one, range(n))); dt = time.perf_counter() - t
    print("threads", T, "tasks", n, "wall ms", dt * 1e3, "per-task-per-thread ms", dt * 1e3 / 8)
