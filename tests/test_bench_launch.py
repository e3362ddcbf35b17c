"""bench.py --gpus N from a bare shell (the driver's command shape): the parent must start torch.distributed.run as a
child process, relay rank 0's JSON line and the exit code.  CPU only: the argument plumbing and
the failure path (no GPU here: the ranks refuse, the parent reports their exit code); the rehearsal on the GPU box is in
tests/test_gpu_sharded.py."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_relaunch_command_is_the_drivers_line():
    b = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = b.relaunch_command(8, argv, port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                       # the ranks see exactly the parent's arguments
    p = b.relaunch_command(2, [])[b.relaunch_command(2, []).index("--master-port") + 1]
    assert 1024 < int(p) < 65536                     # a free port picked by the kernel


def test_bare_shell_launch_relays_the_ranks_exit_code():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--edge", "6", "--steps", "1",
                        "--warmup", "0", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr   # it did launch itself
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0                     # the ranks refused (HIP only, no CPU fallback); nothing on stdout
        assert "needs an MI355X" in r.stderr
        assert r.stdout.strip() == ""


def test_direct_torchrun_form_still_checks_world_size():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-sample", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
