"""N > 1 plumbing on CPU: two processes, gloo backend.  The row-sharded search must return exactly
what one unsharded index returns (ids global, same tie order) on every rank.  The local searcher
and the merge are injected (the oracle stands in for the HIP index and the HIP merge kernel --
test infrastructure only); what is under test is textreact_amd.sharded: shard bounds, id offsets,
the packed all-gather layout, the merge input order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleLocalIndex:
    """stands in for faiss_compat.IndexFlat: add(tensor), search_s64(tensor, k)"""

    def __init__(self, metric):
        self.metric, self.y = metric, None

    def add(self, x):
        x = x.numpy().astype(np.float32)
        self.y = x if self.y is None else np.concatenate([self.y, x])

    ntotal = property(lambda self: 0 if self.y is None else self.y.shape[0])

    def search(self, x, k):
        D, I, _ = self.search_s64(x, k)
        return D, I

    def search_s64(self, x, k):
        from oracle import flat_knn as oracle
        xq = x.numpy().astype(np.float32)
        D, I = oracle.knn_canonical(self.metric, xq, self.y, k)
        S = oracle.scores_at(self.metric, xq, self.y, I)
        pad = np.finfo(np.float32).max * (1 if self.metric == 1 else -1)
        S = np.where(I >= 0, S, pad)
        return torch.from_numpy(D), torch.from_numpy(I), torch.from_numpy(S)


def oracle_merge(metric, S_all, I_all):
    # fp64 scores decide; ties by id -- same rule as trx_merge_topk_device
    S, I = S_all.numpy(), I_all.numpy()
    nl, nq, k = S.shape
    key = S if metric == 1 else -S
    flatS, flatI, flatK = (a.transpose(1, 0, 2).reshape(nq, nl * k) for a in (S, I, key))
    big = np.where(flatI >= 0, flatK, np.inf)
    order = np.lexsort((np.where(flatI >= 0, flatI, np.iinfo(np.int64).max), big), axis=1)[:, :k]
    Iout = np.take_along_axis(flatI, order, 1)
    Sout = np.take_along_axis(flatS, order, 1)
    return torch.from_numpy(Sout.astype(np.float32)), torch.from_numpy(Iout)


def _worker(rank, world, port, metric, n, d, nq, k, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _data import grid
    from textreact_amd.sharded import ShardedFlatIndex, shard_bounds
    y, x = grid(n, d, 1), grid(nq, d, 2)
    lo, hi = shard_bounds(n, world, rank)
    idx = ShardedFlatIndex(d, metric, local_index=OracleLocalIndex(metric), merge=oracle_merge)
    idx.add_shard(torch.from_numpy(y[lo:hi]), lo, n)
    D, I = idx.search(torch.from_numpy(x), k)
    ret[rank] = (D.numpy(), I.numpy())
    dist.destroy_process_group()


def _replica_worker(rank, world, port, metric, n, d, nq, k, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _data import grid
    from textreact_amd.sharded import ReplicatedFlatIndex
    idx = ReplicatedFlatIndex(d, metric, local_index=OracleLocalIndex(metric))
    idx.add(torch.from_numpy(grid(n, d, 1)))
    D, I = idx.search(torch.from_numpy(grid(nq, d, 2)), k)
    ret[rank] = (D.numpy(), I.numpy(), idx.ntotal)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("metric", [0, 1])
def test_two_rank_sharded_search_equals_unsharded(metric):
    from _data import grid
    from oracle import flat_knn as oracle
    n, d, nq, k = 1001, 32, 37, 10   # odd n: uneven shards; grid values: real ties across shards
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), metric, n, d, nq, k, ret), nprocs=2, join=True)
    Dr, Ir = oracle.knn_canonical(metric, grid(nq, d, 2), grid(n, d, 1), k)
    for r in (0, 1):
        D, I = ret[r]
        assert np.array_equal(I, Ir), "rank %d" % r
        assert np.array_equal(D, Dr)


@pytest.mark.parametrize("world,nq", [(2, 37), (3, 2), (3, 100)])
def test_replicated_search_equals_one_index(world, nq):
    """FAISS' IndexReplicas form (every rank all rows, a G-th of the queries each, one all-gather of the slices): uneven
    slices, and fewer queries than ranks (a rank with an empty slice still joins the collective)"""
    from _data import grid
    from oracle import flat_knn as oracle
    n, d, k = 500, 32, 10
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_replica_worker, args=(world, _free_port(), 1, n, d, nq, k, ret), nprocs=world, join=True)
    Dr, Ir = oracle.knn_canonical(1, grid(nq, d, 2), grid(n, d, 1), k)
    for r in range(world):
        D, I, nt = ret[r]
        assert nt == n and np.array_equal(I, Ir) and np.array_equal(D, Dr), "rank %d" % r


def test_shard_bounds_cover_and_balance():
    from textreact_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 1000000, 1000003):
        for g in (1, 2, 3, 8):
            b = [shard_bounds(n, g, r) for r in range(g)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(g - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _grid_worker(rank, world, port, metric, row_groups, n, d, nq, k, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _data import grid
    from textreact_amd.sharded import ShardedFlatIndex
    y, x = grid(n, d, 1), grid(nq, d, 2)
    idx = ShardedFlatIndex(d, metric, local_index=OracleLocalIndex(metric), merge=oracle_merge, row_groups=row_groups)
    lo, hi = idx.shard_rows(n)
    idx.add_shard(torch.from_numpy(y[lo:hi]), lo, n)
    D, I = idx.search(torch.from_numpy(x), k)
    ret[rank] = (D.numpy(), I.numpy(), (idx.row_rank, idx.query_rank, lo, hi))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,row_groups,nq", [(4, 2, 37), (4, 1, 9), (4, 4, 37), (6, 3, 5), (6, 2, 101)])
def test_rows_by_queries_grid_equals_unsharded(world, row_groups, nq):
    """round 6: G = Gr x Gq.  Rank j Gr + i holds row shard i of Gr and searches query slice j of Gq; the candidate exchange runs
    inside a column, the final all-gather over all ranks.  2 x 2, the two degenerate grids (Gr = 1: replicas; Gr = G: pure row
    sharding), 3 x 2 and 2 x 3 with fewer queries than ranks / uneven slices: the unsharded oracle's (D, I) on every rank"""
    from _data import grid
    from oracle import flat_knn as oracle
    from textreact_amd.sharded import shard_bounds
    n, d, k = 1001, 32, 10
    for metric in (0, 1):
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(_grid_worker, args=(world, _free_port(), metric, row_groups, n, d, nq, k, ret), nprocs=world, join=True)
        Dr, Ir = oracle.knn_canonical(metric, grid(nq, d, 2), grid(n, d, 1), k)
        for r in range(world):
            D, I, (i, j, lo, hi) = ret[r]
            assert (i, j) == (r % row_groups, r // row_groups) and (lo, hi) == shard_bounds(n, row_groups, i)
            assert np.array_equal(I, Ir) and np.array_equal(D, Dr), "rank %d of %d x %d" % (r, row_groups, world // row_groups)


def test_grid_and_tie_rule_arguments_are_checked_at_construction():
    from textreact_amd.sharded import ShardedFlatIndex

    class Local(OracleLocalIndex):
        tie_rule = "faiss"

        def set_tie_rule(self, r):
            self.tie_rule = r
    with pytest.raises(TypeError, match="faiss_ties_k"):          # a merge callable without the keyword: found here, not inside a collective
        ShardedFlatIndex(8, 0, local_index=Local(0), merge=oracle_merge)
    loc = Local(0)
    idx = ShardedFlatIndex(8, 0, local_index=loc, merge=lambda m, S, I, faiss_ties_k=None: oracle_merge(m, S, I))
    assert loc.tie_rule == "id" and idx.tie_rule == "faiss"       # the wrapper took the rule over ...
    with pytest.raises(AssertionError, match="1024"):
        idx.search(torch.zeros(1, 8), 1025)
    assert idx.release_local() is loc and loc.tie_rule == "faiss"  # ... and gives it back
    with pytest.raises(ValueError, match="does not divide"):
        ShardedFlatIndex(8, 0, local_index=OracleLocalIndex(0), merge=oracle_merge, row_groups=3)
