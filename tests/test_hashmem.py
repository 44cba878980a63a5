"""usehashtable on a batch WITHOUT a GPU (reference wsprd.c:481-494, 842-852, wsprd_utils.c:264-311,
wsprsim_utils.c:280-300): the product's batch hash memory (HashBatch / SegHashView) and its per-thread message cache
(MessageCache), compiled straight from rtlsdr-wsprd_amd/csrc/host/wspr_hashmem.cpp + wspr_message.cpp, driven by a model
decoder whose behaviour depends on what its type-3 look-ups answer (tests/helpers/hashmem_check.cpp).  Rounds of parallel
"decodes" against logged views must give, text for text, symbol vector for symbol vector and byte for byte of
hashtable.txt, what ONE table walked through the segments in order gives -- as a single call and as shards exchanging
their stores.  The GPU suite checks the same against the oracle's real decodes (tests/test_gpu_hashtable.py)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hm(tmp_path_factory):
    so = tmp_path_factory.mktemp("hm") / "hashmem_check.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-mpopcnt", "-ffp-contract=off", "-shared", "-fPIC", "-Wno-format-truncation",
                    "-o", str(so), os.path.join(ROOT, "tests", "helpers", "hashmem_check.cpp")], check=True)
    L = C.CDLL(str(so))
    L.hashmem_selftest.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_char_p, C.c_int]
    return L


@pytest.mark.parametrize("frac23,nshards", [(0.0, 1), (0.05, 1), (0.5, 1), (0.3, 3), (0.3, 8)])
def test_rounds_against_views_equal_the_serial_walk(hm, tmp_path, frac23, nshards):
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        for seed in (1, 2, 3):
            rounds, redone, resolved = C.c_int(0), C.c_int(0), C.c_int(0)
            err = C.create_string_buffer(256)
            rc = hm.hashmem_selftest(seed * 7919 + int(frac23 * 100) + nshards, 400, 6, frac23, nshards, C.byref(rounds),
                                     C.byref(redone), C.byref(resolved), err, 256)
            assert rc == 0, err.value.decode()
            print("frac23 %.2f, %d shard(s), seed %d: %d extra rounds, %d segments decoded again, %d hashed calls resolved"
                  % (frac23, nshards, seed, rounds.value, redone.value, resolved.value))
            if frac23 == 0.0:
                assert redone.value == 0 and resolved.value == 0
            else:
                assert redone.value > 0 and resolved.value > 0
    finally:
        os.chdir(cwd)
