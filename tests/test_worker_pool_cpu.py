"""CPU: the persistent threads that stage a batch (nimble_amd/host/worker_pool.h)
under ThreadSanitizer -- jobs of changing width, several callers at once, every
task taken exactly once, no report.  (The GPU runtime does not start under
ThreadSanitizer, so the pool is exercised on its own.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_worker_pool_under_thread_sanitizer(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "worker_pool_stress"
    built = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + os.path.join(ROOT, "nimble_amd", "host"),
                            os.path.join(ROOT, "tests", "native", "worker_pool_stress.cpp"), "-lpthread", "-o", str(exe)],
                           capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("ThreadSanitizer build not available here: " + built.stderr[-300:])
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    if run.returncode != 0 and "FATAL: ThreadSanitizer" in run.stderr and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert run.stdout.startswith("ok 850")
