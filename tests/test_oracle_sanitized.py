"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU sanitizers only: the pool has no GPU
sanitizer).  `make -C oracle san` builds oracle/_san/libmrf_oracle_san.so from the same source; the oracle's own test
files -- the golden vectors and the pin / reconciliation plumbing -- then run in a child interpreter with libasan
preloaded and MRF_ORACLE_LIB pointing at that build.  A sanitizer finding aborts the child."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san", "libmrf_oracle_san.so")


def _libasan():
    cxx = os.environ.get("CXX", "g++")
    if not shutil.which(cxx):
        return None
    out = subprocess.run([cxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(_libasan() is None, reason="no g++ / libasan on this host")
def test_oracle_tests_pass_under_asan_and_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    env = dict(os.environ, LD_PRELOAD=_libasan(), MRF_ORACLE_LIB=SAN, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
           os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_reference_pin.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
