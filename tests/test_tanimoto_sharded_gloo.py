"""Row-sharded Tanimoto search on CPU: two processes, gloo.  The local searcher is injected (the numpy oracle forms
the keys the HIP kernels form); under test is ShardedTanimotoIndex: row offsets inside the keys, padding of short shards,
the packed all-gather and the merge -- the result must equal one unsharded search on every rank, tie order included."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLocalTanimoto:
    """stands in for TanimotoIndex: add(fps), search_keys(queries, k) -> (key, and, den)"""

    def __init__(self):
        self.c = None

    def add(self, fps):
        self.c = np.abs(np.asarray(fps)).astype(np.int64)

    def search_keys(self, queries, k, batch=None):
        q = np.abs(np.asarray(queries)).astype(np.int64)
        n = len(self.c)
        kk = min(k, n)
        K, A, D = (np.zeros((len(q), kk), dtype=np.int64) for _ in range(3))
        for i, qi in enumerate(q):
            a = np.minimum(self.c, qi[None]).sum(1)
            den = self.c.sum(1) + qi.sum() - a
            sim = np.where(den > 0, a / np.maximum(den, 1), 0.0)
            keys = ((sim * 68719476735.0).astype(np.int64) << 27) | np.arange(n, dtype=np.int64)
            top = np.argsort(keys)[::-1][:kk]
            K[i], A[i], D[i] = keys[top], a[top], den[top]
        return torch.from_numpy(K), torch.from_numpy(A), torch.from_numpy(D)


def _worker(rank, world, port, n, d, nq, k, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_tanimoto_cpu import fingerprints
    from textreact_amd.sharded import shard_bounds
    from textreact_amd.tanimoto import ShardedTanimotoIndex
    rng = np.random.default_rng(7)
    base = fingerprints(rng, 30, d, density=0.1)
    corpus = base[rng.integers(0, 30, n)]                 # many equal similarities, spread over both shards
    queries = np.concatenate([base[:nq - 1], np.zeros((1, d), dtype=np.int64)])
    lo, hi = shard_bounds(n, world, rank)
    idx = ShardedTanimotoIndex(d, local_index=OracleLocalTanimoto())
    idx.add_shard(corpus[lo:hi], lo, n)
    sim, rank_ = idx.search(queries, k)
    ret[rank] = (sim.numpy(), rank_.numpy(), corpus, queries)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(n, d, nq, k):
    from oracle import tanimoto as oracle
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), n, d, nq, k, ret), nprocs=2, join=True)
    for r in (0, 1):
        sim, rank, corpus, queries = ret[r]
        want_s, want_r = oracle.search(queries, corpus, k)
        assert np.array_equal(rank, want_r), "rank %d" % r
        assert np.array_equal(sim, want_s)


def test_two_rank_sharded_tanimoto_equals_unsharded():
    _run(1001, 64, 9, 100)


def test_shards_shorter_than_k_are_padded():
    _run(151, 64, 5, 100)


def _cli_worker(rank, world, port, argv):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      TRX_DIST_BACKEND="gloo")
    from textreact_amd import tanimoto
    real = tanimoto.retrieve_sharded
    tanimoto.retrieve_sharded = lambda *a, **k: real(*a, local_index=OracleLocalTanimoto(), **k)     # the oracle stands in for the HIP index
    assert tanimoto.main(argv) == 0


def test_two_rank_command_line_writes_the_unsharded_result(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m textreact_amd.tanimoto ...: rows split over the ranks, rank 0 writes
    the structure retrieve.py dumps -- equal to the unsharded oracle's, tie order included"""
    import json
    from oracle import tanimoto as oracle
    from test_tanimoto_cpu import fingerprints
    rng = np.random.default_rng(12)
    base = fingerprints(rng, 30, 64, density=0.1)
    corpus = base[rng.integers(0, 30, 501)]
    queries = base[:7]
    np.save(tmp_path / "train.npy", corpus); np.save(tmp_path / "test.npy", queries)
    argv = ["--train_fps", str(tmp_path / "train.npy"), "--test_fps", str(tmp_path / "test.npy"), "--limit", "-1", "--output", str(tmp_path / "two.json")]
    mp.spawn(_cli_worker, args=(2, _free_port(), argv), nprocs=2, join=True)
    got = json.load(open(tmp_path / "two.json"))
    want_s, want_r = oracle.search(queries, corpus, 100)
    assert sorted(got, key=int) == [str(i) for i in range(7)]
    for i in range(7):
        assert got[str(i)]["rank"] == want_r[i].tolist() and got[str(i)]["similarity"] == want_s[i].tolist()
