"""CPU: the host code that parses untrusted bytes (planner, stream splitter,
decompression pre-pass, file front-end) built with AddressSanitizer +
UndefinedBehaviorSanitizer and driven by tools/plan_fuzz.cpp over a small seed
corpus: the unchanged seeds first, then mutated copies.  Any sanitizer report
or abort fails the test.  (The long campaigns are run by hand with the same
binary; this keeps the harness building and the seeds clean.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sanitized_parsers_survive_mutated_inputs(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "plan_fuzz"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan",
           "-w", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "nimble_amd", "host"),
           os.path.join(ROOT, "tools", "plan_fuzz.cpp"), os.path.join(ROOT, "nimble_amd", "csrc", "nbl_plan.cpp"),
           os.path.join(ROOT, "nimble_amd", "csrc", "nbl_inflate.cpp"), os.path.join(ROOT, "nimble_amd", "host", "tablet.cpp"),
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-lpthread",
           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("sanitizer build not available here: " + built.stderr[-300:])
    corpus = tmp_path / "corpus.bin"
    made = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_corpus.py"), str(corpus), "120"],
                          capture_output=True, text=True)
    assert made.returncode == 0, made.stderr[-2000:]
    run = subprocess.run([str(exe), str(corpus), "30000", "5"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "done: planned" in run.stdout
