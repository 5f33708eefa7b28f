"""bench.py's launch contract, the parts that are decided before anything touches a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(kw)
    return env


def test_gpus_that_disagree_with_the_launcher_are_refused():
    """--gpus 2 under a launcher that set WORLD_SIZE=3 (or 1) exits non-zero before it imports torch: no bench line whose
    n_gpus differs from the command's"""
    for ws in ("3", "1"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                           timeout=120, env=_env(WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0"))
        assert r.returncode == 2 and "refusing" in r.stderr, (r.returncode, r.stderr[-500:])
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_without_a_launcher_starts_one_as_a_child(tmp_path):
    """--gpus 2 and no WORLD_SIZE: bench.py starts torch.distributed.run itself and exits with its code.  No GPU here, so
    every rank stops at 'bench.py needs a GPU' -- what is checked is that two ranks were started and the code relayed."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr, r.stderr[-1500:]
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stderr, r.stderr[-1500:]


def test_traffic_is_reported_only_for_the_kernel_text_it_was_measured_on(tmp_path):
    import bench
    root = tmp_path
    for rel in bench.SCAN_SOURCES:
        os.makedirs(os.path.dirname(root / rel), exist_ok=True)
        (root / rel).write_text("// kernel text of " + rel)
    os.makedirs(root / "profiles")
    sha = bench.scan_source_sha256(str(root))
    (root / "profiles" / "traffic.json").write_text(json.dumps(
        {"git_sha": "abc1234", "scan_source_sha256": sha, "hbm_bytes_per_launch": 1.5e11, "avg_launch_ms_kernel_trace": 70.0}))
    t, src = bench.committed_traffic(str(root))
    assert t == 1.5e11 and "abc1234" in src
    (root / bench.SCAN_SOURCES[0]).write_text("// a later kernel")
    t, src = bench.committed_traffic(str(root))
    assert t is None and "STALE" in src
    (root / "profiles" / "traffic.json").write_text(json.dumps({"git_sha": "abc1234", "hbm_bytes_per_launch": 1.5e11}))
    assert bench.committed_traffic(str(root))[0] is None        # a file from before the hash existed is stale by definition
