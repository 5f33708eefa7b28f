"""textreact_amd/_dist.py: one group setup, one refusal rule for every N > 1 entry point (the reference hands `--gpus` to
Lightning's DDP strategy, main.py:372-374).  CPU-only: what must happen BEFORE a rank touches a GPU or a process group."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRX_DEVICE", "TRX_DIST_BACKEND")}
    env.update(kw); env["PYTHONPATH"] = ROOT
    return env


def test_refuse_mismatch_rule(monkeypatch):
    from textreact_amd import _dist
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    _dist.refuse_mismatch(4, "x")                       # no launcher: nothing to disagree with
    monkeypatch.setenv("WORLD_SIZE", "4")
    _dist.refuse_mismatch(4, "x")
    _dist.refuse_mismatch(None, "x")                    # a CLI without the flag
    with pytest.raises(SystemExit) as e:
        _dist.refuse_mismatch(2, "x")
    assert e.value.code == 2


def test_device_ordinal_and_world(monkeypatch):
    from textreact_amd import _dist
    for k in ("WORLD_SIZE", "LOCAL_RANK", "TRX_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    assert _dist.world_size() == 1 and _dist.device_ordinal() == 0
    monkeypatch.setenv("LOCAL_RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "8")
    assert _dist.world_size() == 8 and _dist.device_ordinal() == 3
    monkeypatch.setenv("TRX_DEVICE", "0")               # rehearsal: several ranks on one GPU
    assert _dist.device_ordinal() == 0


def test_setup_without_a_launcher_makes_no_group():
    import torch.distributed as dist
    from textreact_amd import _dist
    old = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        rank, world, device = _dist.setup()
        assert (rank, world) == (0, 1) and not dist.is_initialized()
    finally:
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


def test_main_refuses_a_gpus_flag_that_is_not_the_world_size(tmp_path):
    """main.py --gpus 2 under a launcher that set WORLD_SIZE=3: exit code 2 and the reason, before any GPU or group call"""
    r = subprocess.run([sys.executable, "-m", "textreact_amd.main", "--gpus", "2", "--save_path", str(tmp_path)], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_env(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and "--gpus 2" in r.stderr, r.stderr[-2000:]


def test_bench_refuses_the_same_way():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       env=_env(WORLD_SIZE="3"))
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr and r.stdout.strip() == ""


@pytest.mark.parametrize("module", ["textreact_amd.retrieve_faiss", "textreact_amd.tanimoto"])
def test_rccl_ranks_without_a_gpu_each_are_refused_with_the_reason(module, tmp_path):
    """the two retrieval CLIs have no --gpus flag (the reference's have none): what they refuse is a rank of an RCCL group
    that has no GPU of its own -- here: local rank 63 of a box with fewer GPUs -- in _dist.setup, exit code 2, not a hang inside the first collective"""
    import numpy as np
    np.save(tmp_path / "a.npy", np.zeros((4, 8), np.int8))
    if module.endswith("tanimoto"):
        argv = ["--train_fps", str(tmp_path / "a.npy"), "--test_fps", str(tmp_path / "a.npy"), "--output", str(tmp_path / "o.json")]
    else:
        import pandas as pd
        pd.DataFrame({"id": ["a", "b", "c", "d"]}).to_csv(tmp_path / "t.csv", index=False)
        argv = ["--data_path", str(tmp_path), "--train_file", "t.csv", "--valid_file", "t.csv", "--test_file", "t.csv", "--output_path", str(tmp_path / "out"),
                "--train_vectors", str(tmp_path / "a.npy"), "--valid_vectors", str(tmp_path / "a.npy"), "--test_vectors", str(tmp_path / "a.npy")]
    r = subprocess.run([sys.executable, "-m", module] + argv, capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=_env(WORLD_SIZE="64", RANK="63", LOCAL_RANK="63", TRX_DIST_BACKEND="nccl", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    # (local rank 63: refused on this box whether it has no GPU or eight)
    assert r.returncode == 2 and "one GPU per" in r.stderr and "cuda:63" in r.stderr, (r.returncode, r.stderr[-1500:])


def _setup_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    os.environ.pop("TRX_DIST_BACKEND", None)
    import torch.distributed as dist
    from textreact_amd import _dist
    r, w, dev = _dist.setup()                 # no GPU here: the gloo backend, device cpu
    again = _dist.setup()                     # a second call finds the group and changes nothing
    import torch
    t = torch.tensor([r + 1.0]); dist.all_reduce(t)
    ret[rank] = (r, w, str(dev), again[:2] == (r, w), float(t), dist.get_backend())
    dist.destroy_process_group()


def test_setup_makes_the_group_once_for_every_rank():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_setup_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in (0, 1):
        assert ret[r] == (r, 2, "cpu", True, 3.0, "gloo"), ret[r]
