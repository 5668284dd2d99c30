"""librmn_amd/csrc/libm_exact.h (the C library's REAL sinf / cosf / asinf / atanf / atan2f restated for the device locate of rotated sources: ez_lac.inc:31-47,
ez_cal.inc:22-47) compiled for the host and compared with this machine's C library, bit for bit: every 251st REAL bit pattern (17 M arguments per function) and
8 M pairs here; tools/check_libm_exact.c without a stride covers all 2^32 (run once per round, profiles/r05_experiments.txt).  The device compilation of the
same header is compared in tests/test_gpu_interp.py::test_libm_exact_on_the_device_equals_the_c_library."""
import os, subprocess
import pytest
from conftest import both_legs

_R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


@both_legs
def test_libm_exact_header_equals_the_c_library(leg, tmp_path):
    exe = str(tmp_path / "check_libm_exact")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-pthread", os.path.join(_R, "tools", "check_libm_exact.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe, "8", "8", "251"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sinf 0, cosf 0, asinf 0, atanf 0" in out.stdout and "atan2f 0 of" in out.stdout, out.stdout


@both_legs
def test_library_recognises_this_c_library(leg):
    """the library's own sampled check (made once, before the first rotated source is located on the device) agrees with the exhaustive one above on this image"""
    import ctypes, time
    from librmn_amd import lib
    L = lib.load_library()
    t0 = time.perf_counter()
    assert L.ezhip_libm_exact_matches_host() == 1
    assert time.perf_counter() - t0 < 0.5
