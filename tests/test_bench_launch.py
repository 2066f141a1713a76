"""CPU: `python bench.py --gpus N` starts its own ranks: the parent never touches the GPU, the children are a
torch.distributed.run process tree, rank 0's JSON line comes through, the exit code is the children's."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""  # plumbing only, even on a GPU box
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["dry_launch"] and out["n_gpus"] == 2 and out["config"] == {"ranks": 2, "backend": "gloo"} and out["max_over_ranks"] == 2.0
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr


def test_bench_under_the_drivers_launcher():
    """The driver's own command line for N > 1 (RANK / WORLD_SIZE already set): bench.py is a rank, it must not launch again."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["config"]["ranks"] == 2
    assert "starting 2 ranks" not in r.stderr


def test_child_failure_is_the_parents_exit_code():
    """Without a ROCm device the ranks refuse to run (no CPU fallback): the parent relays the failure."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a ROCm device" in r.stderr
