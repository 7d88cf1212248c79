"""bench.py's launcher contract, on CPU: `python bench.py --gpus N` with no launcher around it must start N ranks
itself (the driver's single-process command shape), `--gpus N` under a launcher of a different size must refuse,
and the JSON line must be the last line of stdout with n_gpus = the size of the process group that actually ran."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=env, timeout=timeout, cwd=ROOT)


def test_plain_python_gpus2_self_launches_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    out = json.loads(last)
    assert out == {"launch_check": True, "n_gpus": 2, "backend": "gloo", "scaling": "weak", "per_rank_batch": 16,
                   "global_batch": 32, "min_kept": 16 * 1024 * 1024 // 16}


def test_strong_scaling_splits_the_global_batch_like_the_reference():
    """--scaling strong: the reference's setting (bisenet dataloader.py:51-53 batch_size // world_size; train.py:48-49
    min_kept from the per-rank batch): global 16 -> 8 per rank at 2 ranks, min_kept 8 * 1024^2 / 16; a split that leaves a
    rank fewer than 2 images is refused."""
    r = _run(["--gpus", "2", "--launch-check", "--scaling", "strong"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["scaling"] == "strong" and out["per_rank_batch"] == 8 and out["global_batch"] == 16
    assert out["min_kept"] == 8 * 1024 * 1024 // 16
    r = _run(["--gpus", "1", "--launch-check", "--scaling", "strong", "--batch", "1"])
    assert r.returncode != 0 and "strong" in r.stderr


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in r.stderr


def test_defaults_follow_survey_8d():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--steps", type=int, default=50' in src and '"--warmup", type=int, default=20' in src
