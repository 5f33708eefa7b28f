"""Row-sharded exact k-NN across the GPUs of one node.

One process per GPU.  Rank r holds corpus rows [r*N/G, (r+1)*N/G) in its own flat index; queries are
replicated.  A search is: local search -> (fp64 score, local id + shard offset) -> ONE all-gather
of nq*k*16 bytes per rank (RCCL over xGMI: `torch.distributed` backend "nccl") -> every rank merges
the G lists with the same total order as a single unsharded index, so the result is identical to
an unsharded IndexFlat over the concatenated corpus.

The reference has no multi-GPU retrieval (retrieve/retrieve_faiss.py is single-process CPU FAISS);
this is the north-star's scale-out of index_and_search (retrieve_faiss.py:62-74).

`local_index` and `merge` are injectable so the N > 1 plumbing (shard bounds, id offsets, gather
layout, merge order) is covered on CPU with the gloo backend in tests/test_sharded_gloo.py, where
the test supplies the oracle for both; the defaults are the HIP index and the HIP merge kernel.
"""
from typing import Callable, Optional, Tuple


MAX_SHARDS = 16   # nlists limit of trx_merge_topk_device (csrc/knn_api.hip)


def shard_bounds(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row ranges: the first n_total % G shards get one extra row."""
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedFlatIndex:
    def __init__(self, d: int, metric: int, group=None, local_index=None, merge: Optional[Callable] = None):
        import torch.distributed as dist
        self.d, self.metric = int(d), int(metric)
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world_size > MAX_SHARDS:
            raise ValueError("ShardedFlatIndex: %d ranks, but trx_merge_topk_device merges at most %d lists "
                             "(include/trx_knn.h); shard over the GPUs of one node" % (self.world_size, MAX_SHARDS))
        if local_index is None:
            from . import faiss_compat
            local_index = faiss_compat.IndexFlat(d, metric)
        if merge is None:
            from . import faiss_compat
            merge = faiss_compat.merge_topk
        self.local = local_index
        self._merge = merge
        self.offset = 0      # global id of this shard's first row
        self.ntotal = 0      # rows over all shards

    def add_shard(self, x_local, offset: int, ntotal: int):
        """Add this rank's rows (already partitioned by the caller, e.g. with shard_bounds)."""
        self.local.add(x_local)
        self.offset, self.ntotal = int(offset), int(ntotal)

    def search(self, x, k: int):
        """x replicated on every rank (torch tensor on this rank's device).  Returns (D, I) with
        global ids, identical on every rank."""
        import torch
        import torch.distributed as dist
        D, I, S = self.local.search_s64(x, k)
        I = torch.where(I >= 0, I + self.offset, I)
        if self.world_size == 1:
            return D, I
        # one collective: pack (fp64 score bits, id) as int64 [2, nq, k]; nq*k*16 bytes per rank
        nq = S.shape[0]
        pack = torch.stack([S.contiguous().view(torch.int64), I.contiguous()])
        out = torch.empty((self.world_size,) + tuple(pack.shape), dtype=torch.int64, device=pack.device)
        if dist.get_backend(self.group) == "gloo":  # test path (gloo gathers host tensors only)
            host = pack.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world_size)]
            dist.all_gather(parts, host, group=self.group)
            out = torch.stack(parts).to(pack.device)
        else:
            dist.all_gather_into_tensor(out, pack, group=self.group)
        S_all = out[:, 0].contiguous().view(torch.float64)
        I_all = out[:, 1].contiguous()
        return self._merge(self.metric, S_all, I_all)
