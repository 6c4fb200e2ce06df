"""bench.py --gpus N without a launcher starts the N ranks itself -- from a process that has not imported torch or the
GPU library (a process that has initialised HIP must never become a launcher on the GPU pool)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra, timeout=240):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LB_BENCH_CHILD"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


def test_launcher_command_is_built_before_anything_touches_the_gpu():
    r = run(["--gpus", "8", "--steps", "3", "--warmup", "1"], {"LB_BENCH_LAUNCH_ECHO": "1"})
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["torch_imported"] is False and info["longbow_imported"] is False
    cmd = info["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]  # the ranks get the caller's arguments


def test_launcher_relays_the_childs_failure():
    """no GPU here: every rank fails (GPU not available / no HIP device) and the launcher exits non-zero instead of
    printing a line"""
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-legs", "--no-cpu-baseline"], {})
    assert r.returncode != 0
    assert not any(l.startswith("{") and '"metric"' in l for l in r.stdout.splitlines())


def test_a_rank_under_an_external_launcher_does_not_spawn_again():
    """with WORLD_SIZE set (torch.distributed.run started us) the process is a rank: it must go on to the GPU set-up,
    which fails here for lack of a device -- but not by recursing into the launcher"""
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                                 "LB_BENCH_LAUNCH_ECHO": "1", "MASTER_ADDR": "127.0.0.1",
                                                                 "MASTER_PORT": "29999"}, timeout=120)
    assert '"cmd"' not in r.stdout
    assert r.returncode != 0
