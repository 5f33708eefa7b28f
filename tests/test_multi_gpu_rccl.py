"""The N > 1 path over real RCCL, one GPU per rank.  Every box this project has been given so far has ONE GPU, so these tests
have never run; they skip themselves there and are what a multi-GPU box runs first."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpus():
    try:
        import torch
        return torch.cuda.device_count()          # (does not initialise the GPU)
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_ngpus() < 2, reason="one GPU on this box: RCCL refuses two ranks on one device")


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@needs2
@pytest.mark.parametrize("mode", [[], ["--replicas"], ["--weak"]])
def test_bench_over_rccl(mode):
    """`python bench.py --gpus G` (no launcher: it starts its own ranks), G = 2 and every GPU of the box, small sizes: the
    row-sharded line checks its exchange against an independent computation before it times it"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRX_BENCH_BACKEND", "TRX_BENCH_DEVICE")}
    for g in sorted({2, min(_ngpus(), 8)}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(g), "--steps", "2", "--warmup", "1",
                            "--n-corpus", "60000", "--n-queries", "2048"] + mode, capture_output=True, text=True, timeout=1200, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        j = _line(r.stdout)
        assert j["n_gpus"] == g and j["value"] > 0
        assert "RCCL" in j["config"]["transport"]
        if "--replicas" not in mode:
            assert j["config"]["selfcheck"].startswith("passed"), j["config"]


def _rank(rank, world, port, ret):
    import torch, torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import textreact_amd.faiss_compat as fc
    from textreact_amd.sharded import ShardedFlatIndex, shard_bounds
    from _data import gaussian, grid
    out = {}
    y, x = gaussian(30011, 96, 11), gaussian(301, 96, 12)
    y[5000:5040] = y[100]; y[20000:20020] = y[100]                       # ties that straddle shard boundaries
    lo, hi = shard_bounds(len(y), world, rank)
    for metric in (0, 1):
        idx = ShardedFlatIndex(96, metric, local_index=fc.IndexFlat(96, metric, device=rank))
        idx.add_shard(torch.from_numpy(y[lo:hi]).cuda(rank), lo, len(y))
        D, I = idx.search(torch.from_numpy(x).cuda(rank), 10)
        out[metric] = (D.cpu().numpy(), I.cpu().numpy())
    yt, xt = grid(20003, 32, 31), grid(200, 32, 32)
    lo, hi = shard_bounds(len(yt), world, rank)
    idx = ShardedFlatIndex(32, 0, local_index=fc.IndexFlatIP(32, device=rank, tie_rule="faiss"))
    idx.add_shard(torch.from_numpy(yt[lo:hi]).cuda(rank), lo, len(yt))
    D, I = idx.search(torch.from_numpy(xt).cuda(rank), 10)
    out["faiss_ties"] = (D.cpu().numpy(), I.cpu().numpy())
    ret[rank] = out
    dist.destroy_process_group()


@needs2
def test_sharded_search_over_rccl_equals_the_oracle():
    import socket
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _data import gaussian, grid
    from oracle import flat_knn as oracle
    world = min(_ngpus(), 4)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_rank, args=(world, port, ret), nprocs=world, join=True)
    y, x = gaussian(30011, 96, 11), gaussian(301, 96, 12)
    y[5000:5040] = y[100]; y[20000:20020] = y[100]
    for metric in (0, 1):
        Dr, Ir = oracle.knn_canonical(metric, x, y, 10)
        for r in range(world):
            D, I = ret[r][metric]
            assert np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), (metric, r)
    Df, If = oracle.knn_faiss(0, grid(200, 32, 32), grid(20003, 32, 31), 10)
    for r in range(world):
        D, I = ret[r]["faiss_ties"]
        assert np.array_equal(I, If) and np.array_equal(D.view(np.uint32), Df.view(np.uint32)), r
