"""CPU: the C oracle's golden tests replayed through its AddressSanitizer + UBSan build (``make -C oracle asan``; SURVEY.md 5, "race
detection / sanitizers").  MC33's case tables, the fps / ball-query / k-NN loops and the GGM taps index with hand-computed offsets: a read
or write past a buffer, or signed overflow / misaligned access, aborts the child (-fno-sanitize-recover=all).  The sanitized library is
loaded in a child interpreter started with libasan preloaded; the child runs tests/test_oracle_golden.py unchanged (GN_ORACLE_LIB selects
the library oracle.lib() opens)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        path = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception:
        return None
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.skipif(_libasan() is None, reason="gcc's libasan.so is not on this machine")
def test_oracle_goldens_under_address_and_ub_sanitizers():
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(REPO, "oracle", "libgn_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=_libasan(), GN_ORACLE_LIB=so, PYTHONPATH=REPO,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    # the child must really be running the sanitized library
    probe = subprocess.run([sys.executable, "-c", "import oracle, numpy as np; L = oracle.lib(); print(L._name); "
                            "import ctypes; print(hasattr(ctypes.CDLL(None), '__asan_init'))"], env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0, probe.stderr[-2000:]
    assert probe.stdout.split()[0] == so and probe.stdout.split()[1] == "True", probe.stdout
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in out.stdout
