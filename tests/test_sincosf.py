"""The device sinf/cosf (csrc/kernels/glibc_sincosf.h), compiled here for the host, returns the
same float as the host libm for every input the decoder can produce (and, run exhaustively by
hand, for all 2^32 inputs: 0 mismatches with WSPR_SINCOS_FMA=1 on an FMA3 x86-64 host)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    so = tmp_path_factory.mktemp("sc") / "sincosf_check.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so),
                    os.path.join(ROOT, "tests", "helpers", "sincosf_check.cpp"), "-lpthread"], check=True)
    L = C.CDLL(str(so))
    L.sincosf_mismatches.restype = C.c_long
    return L


def _fma_host():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(not _fma_host(), reason="host libm uses the non-FMA variant (34 known 1-ulp differences)")
def test_device_sincosf_equals_host_libm(chk):
    fb = C.c_uint32()
    nthr = min(8, os.cpu_count() or 1)
    # phasor seeds |x| < 2 rad: every float in [2^-13, 2) -- exhaustive
    assert chk.sincosf_mismatches(0x39000000, 0x40000000, 1, nthr, C.byref(fb)) == 0, hex(fb.value)
    # subtraction phases up to ~1e5 rad (fast and table-driven reductions): every 7th float in [2, 2^20)
    assert chk.sincosf_mismatches(0x40000000, 0x49800000, 7, nthr, C.byref(fb)) == 0, hex(fb.value)
    # everything else incl. subnormals, huge arguments, inf/nan: every 1021st bit pattern
    assert chk.sincosf_mismatches(0, 0x7fffffff, 1021, nthr, C.byref(fb)) == 0, hex(fb.value)
