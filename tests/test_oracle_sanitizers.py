"""Host sanitizers: the CPU oracle built with gcc -fsanitize=address,undefined (`make -C oracle asan`) must run
the known-answer tests the reference holds, the golden traces and the More-Thuente branch workloads without a report.  The sanitized library is
loaded by a CHILD interpreter that has the sanitizer runtime preloaded (ASan has to be the first library of a process); leak checking is off
(CPython does not free everything at exit), everything else aborts the child.  CPU only: nothing here touches the GPU or the product library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    cp = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True)
    path = cp.stdout.strip()
    return path if cp.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_suites_are_clean_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan.so on this host")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-B", "asan"], stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "oracle", "libqn_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, QN_ORACLE_LIB=lib, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    suites = ["tests/test_oracle_kat.py", "tests/test_oracle_golden.py", "tests/test_oracle_mt_cases.py"]
    cp = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + suites,
                        cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (cp.stdout + cp.stderr)[-3000:]
    assert cp.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in cp.stdout
    # the child did load the sanitized build (a silent fall-back to the plain library would make this test vacuous)
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import qn_oracle as q; q.lib(); "
                            "print(any('libqn_oracle_asan.so' in l for l in open('/proc/self/maps')))" % ROOT],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr
