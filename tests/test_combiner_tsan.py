"""SearchCombiner (longbow_amd/csrc/lb_combine.h: concurrent single-query searches answered by one batched search) under
ThreadSanitizer on the CPU: twelve threads against a stub "device", requests with two values of k, poisoned requests that fail
their batch.  Every caller gets its own results and return code, batches never mix k, a lone caller is not delayed, and the
sanitizer reports nothing."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_combiner_under_thread_sanitizer(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = tmp_path / "combiner_tsan"
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", os.path.join(ROOT, "tests", "cpp", "combiner_tsan.cpp"),
                        "-I" + os.path.join(ROOT, "longbow_amd", "csrc"), "-lpthread", "-o", str(exe)], capture_output=True, text=True)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower():
        pytest.skip("libtsan not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    for _ in range(3):
        r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
        out = r.stdout + r.stderr
        assert "ThreadSanitizer" not in out, out[-3000:]
        assert r.returncode == 0 and out.strip().endswith("OK"), out[-1000:]
