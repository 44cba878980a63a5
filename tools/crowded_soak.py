import sys, os, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import importlib.util
import oracle_lib as ol
spec = importlib.util.spec_from_file_location('tp','tests/test_gpu_parity.py'); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
import trace_parity
import importlib
w = importlib.import_module('rtlsdr_wsprd_amd')
I, Q = tp.crowded_scenes(48, seed=4321)
for o in ({}, {"npasses": 3}, {"subtraction": 0}, {"quickmode": 1}):
    t = time.time()
    tot, und = trace_parity.check(I, Q, w, ol, o, name=str(o))
    print("crowded trace parity", o, "candidates", tot, "undecoded", und, "%.1fs" % (time.time() - t), flush=True)
