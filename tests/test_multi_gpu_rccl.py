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


def _launch(world, module, argv, port):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRX_DIST_BACKEND", "TRX_DEVICE")}
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                           "--master-port", str(port), "-m", module] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


@needs2
def test_retrieve_faiss_cli_over_rccl_writes_the_files_one_gpu_writes(tmp_path):
    """python -m torch.distributed.run --nproc-per-node G -m textreact_amd.retrieve_faiss (retrieve/retrieve_faiss.py:112-130), the train
    vectors row-sharded over G GPUs, RCCL transport, group set up by textreact_amd/_dist.py: rank 0's three files are byte for byte
    those of the one-GPU run -- the sharded and the replicated form"""
    import pandas as pd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _data import reaction_fp_like
    import textreact_amd.retrieve_faiss as rf
    world = min(_ngpus(), 4)
    fps = reaction_fp_like(1101, 2048, 3)
    fps[700:720] = fps[3]; fps[40:45] = fps[3]                     # ties that straddle shard boundaries
    pd.DataFrame({"id": np.arange(1001), "canonical_rxn": ["C>>C"] * 1001}).to_csv(tmp_path / "train.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 5000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "val.csv", index=False)
    pd.DataFrame({"id": np.arange(50) + 6000, "canonical_rxn": ["C>>C"] * 50}).to_csv(tmp_path / "test.csv", index=False)
    for name, sl in (("train", slice(0, 1001)), ("val", slice(1001, 1051)), ("test", slice(1051, 1101))):
        np.save(tmp_path / (name + ".npy"), fps[sl].astype(np.int64))
    argv = ["--data_path", str(tmp_path), "--train_file", "train.csv", "--valid_file", "val.csv", "--test_file", "test.csv",
            "--train_vectors", str(tmp_path / "train.npy"), "--valid_vectors", str(tmp_path / "val.npy"), "--test_vectors", str(tmp_path / "test.npy")]
    one = tmp_path / "one"
    assert rf.main(argv + ["--output_path", str(one)]) == 0
    for case, extra in enumerate(([], ["--replicas"])):
        many = tmp_path / ("many%d" % case)
        r = _launch(world, "textreact_amd.retrieve_faiss", argv + extra + ["--output_path", str(many)], 29741 + case)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        for name in ("train.json", "val.json", "test.json"):
            assert (many / name).read_bytes() == (one / name).read_bytes(), (extra, name)


@needs2
def test_tanimoto_cli_over_rccl_writes_the_unsharded_result(tmp_path):
    """python -m torch.distributed.run --nproc-per-node G -m textreact_amd.tanimoto (retrieve/retrieve.py:55-66 row-sharded): rank 0's
    file equals the one-GPU run's"""
    import textreact_amd.tanimoto as tanimoto
    world = min(_ngpus(), 4)
    rng = np.random.default_rng(12)
    base = (rng.random((40, 2048)) < 0.03) * rng.integers(-3, 4, (40, 2048))
    corpus = base[rng.integers(0, 40, 3001)]
    queries = base[:9]
    np.save(tmp_path / "train.npy", corpus); np.save(tmp_path / "test.npy", queries)
    argv = ["--train_fps", str(tmp_path / "train.npy"), "--test_fps", str(tmp_path / "test.npy"), "--limit", "-1"]
    assert tanimoto.main(argv + ["--output", str(tmp_path / "one.json")]) == 0
    r = _launch(world, "textreact_amd.tanimoto", argv + ["--output", str(tmp_path / "many.json")], 29751)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert json.load(open(tmp_path / "many.json")) == json.load(open(tmp_path / "one.json"))
