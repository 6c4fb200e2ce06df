"""Build liblongbow_gpu.so for gfx950 with hipcc (in-tree, no torch involved).

    python -m longbow_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblongbow_gpu.so")
OUT_DIAG = os.path.join(HERE, "liblongbow_gpu_diag.so")
SOURCES = ["index.hip", "pq.hip", "comm.hip", "flight.hip", "kernels_gemm.hip", "kernels_gemm_narrow.hip", "kernels_gemm_tall2.hip", "kernels_gemm_tall16.hip", "kernels_scan.hip", "kernels_select.hip", "kernels_finish.hip",
           "kernels_pq.hip", "kernels_pq2.hip", "kernels_filter.hip"]
HEADERS = ["lb_device.h", "lb_host.h", "lb_combine.h", "lb_select.h", os.path.join("..", "..", "include", "longbow_gpu.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, diag=False):
    """diag=True builds liblongbow_gpu_diag.so with -DLB_DIAG: LB_* environment tunables, timing-only
    ablation kernels and the clock probe (tools/ only; select it with LB_GPU_SO=<path>)."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    out = OUT_DIAG if diag else OUT
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".diag.o" if diag else ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + (["-DLB_DIAG"] if diag else []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(out, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
