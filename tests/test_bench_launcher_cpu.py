"""bench.py's own N>1 launcher (`python bench.py --gpus N` typed as such, no torchrun in front),
rehearsed on CPU: the parent spawns the ranks before importing torch, the ranks rendezvous on
127.0.0.1 over gloo, rank 0 prints one JSON line, the exit status is relayed.  The conv submit
itself needs a GPU and is not reached (--rehearse-launch stops right after the process group)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ, DFX_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env,
                          capture_output=True, text=True, timeout=300)


def test_bench_gpus2_self_launch_reaches_process_group():
    p = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--rehearse-launch"])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-1500:], p.stderr[-1500:])
    out = json.loads(lines[0])
    assert out == {"rehearsal": True, "n_gpus": 2, "rank_sum": 1.0, "backend": "gloo"}


def test_bench_world_size_mismatch_fails_before_touching_the_gpu():
    p = _run(["--gpus", "8", "--rehearse-launch"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE=2 but --gpus 8" in p.stderr
