"""bench.py's launch contract without a GPU: `python bench.py --gpus N` must start its own N ranks when it was not
started by torch.distributed.run (the driver launches N = 1 plainly and N > 1 through torch.distributed.run; both must
work), every rank must end up with rank 0's settings blob, and --strong must split BASELINE configs[2]'s 512 frames."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    if env:
        e.update(env)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=e, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--batch", "100"])
    assert out["world_size"] == 2 and out["world_size_seen_by_backend"] == 2
    assert out["frames_per_step"] == 200 and out["scaling"] == "weak"
    assert len(set(out["settings_blob_crc32_per_rank"])) == 1, "ranks disagree about the settings blob"
    assert out["noise_after_broadcast"] == 24          # rank 0's value won


def test_bench_under_torch_distributed_run_and_strong_scaling():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29731", "bench.py", "--gpus", "2", "--dry-run", "--strong", "512"])
    assert out["world_size"] == 2 and out["frames_per_step"] == 512 and out["scaling"] == "strong"


def test_bench_single_process_dry_run():
    out = _run([sys.executable, "bench.py", "--dry-run"])
    assert out["world_size"] == 1 and out["frames_per_step"] == 4096
