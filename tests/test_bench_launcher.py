"""`python bench.py --gpus N` without an external launcher must start N fresh rank processes itself (before any GPU
call) and forward rank 0's JSON line — the form bench.py's docstring documents.  Exercised on the CPU through
`--selftest-launch`: the children rendezvous over gloo on 127.0.0.1, reduce a value and rank 0 prints one line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--selftest-launch"], cwd=ROOT,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    d = _run(2)
    assert d == {"selftest_launch": True, "n_gpus": 2, "max_over_ranks": 2.0}


def test_single_rank_needs_no_launcher():
    assert _run(1)["n_gpus"] == 1


def test_launch_cmd_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_cmd(4, ["--gpus", "4", "--steps", "2"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
