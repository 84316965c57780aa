"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer: the golden-vector, episode, property and envgen suites run
again in a subprocess against oracle/libhns_oracle_asan.so (libasan preloaded).  An out-of-bounds access, a use of an
uninitialised slot caught by UBSan, signed overflow or a misaligned access aborts that process and fails this test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_suites_pass_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libhns_oracle_asan.so"])
    env = dict(os.environ, HNS_ORACLE_SANITIZE="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", OMP_NUM_THREADS="1")
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path[:0] = ['oracle', '.']; import hns_oracle as O; print(O.lib()._name)"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("libhns_oracle_asan.so"), probe.stdout + probe.stderr   # the instrumented build is the one loaded
    suites = ["tests/test_oracle_golden.py", "tests/test_oracle_episode.py", "tests/test_oracle_properties.py", "tests/test_hover_golden.py",
              "tests/test_two_evaders.py", "tests/test_envgen.py", "tests/test_integrator_ode.py"]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *suites],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in out.stdout
