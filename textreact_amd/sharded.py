"""Row-sharded exact k-NN across the GPUs of one node.

One process per GPU.  Rank r holds corpus rows [r*N/G, (r+1)*N/G) in its own flat index; queries are
replicated.  A search is: local search -> (fp64 score, local id + shard offset) -> an all-to-all that hands
rank r every rank's lists for ITS nq/G queries -> rank r merges them with the same total order as a single
unsharded index -> an all-gather of the merged slices (RCCL over xGMI: `torch.distributed` backend "nccl"), so
the result is identical to an unsharded IndexFlat over the concatenated corpus, on every rank.

The reference has no multi-GPU retrieval (retrieve/retrieve_faiss.py is single-process CPU FAISS);
this is the north-star's scale-out of index_and_search (retrieve_faiss.py:62-74).

`local_index` and `merge` are injectable so the N > 1 plumbing (shard bounds, id offsets, gather
layout, merge order) is covered on CPU with the gloo backend in tests/test_sharded_gloo.py, where
the test supplies the oracle for both; the defaults are the HIP index and the HIP merge kernel.
"""
from typing import Callable, Optional, Tuple


MAX_SHARDS = 16   # nlists limit of trx_merge_topk_device (csrc/knn_api.hip)
MAX_K = 2048      # TRX_MAX_K (include/trx_knn.h)


def shard_bounds(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row ranges: the first n_total % G shards get one extra row."""
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedFlatIndex:
    """`row_groups` (round 6): a rows x queries GRID over the G ranks.  G = Gr x Gq; rank r = j * Gr + i holds row shard i of Gr
    (`shard_rows`) and searches query slice j of Gq; the candidate all-to-all and the merge run inside a column (the Gr ranks
    that share a query slice, contiguous ranks), the final all-gather over all G.  Gr = G (the default, `row_groups=None`) is
    the north-star form: pure row sharding, every rank searches every query.  A shard's fixed costs -- the per-(query tile,
    corpus split) warm-up of the scan, the select pass -- scale with the queries a rank searches, so at 8 ranks 2 x 4 searches
    500,000 rows x 16,384 queries per rank instead of 125,000 x 65,536; the corpus then has to fit Gr GPUs."""

    def __init__(self, d: int, metric: int, group=None, local_index=None, merge: Optional[Callable] = None,
                 exchange_always: bool = False, row_groups: Optional[int] = None, stream_ordered: bool = True):
        import torch.distributed as dist
        self.d, self.metric = int(d), int(metric)
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.row_groups = self.world_size if row_groups is None else int(row_groups)
        if self.row_groups < 1 or self.world_size % self.row_groups:
            raise ValueError("ShardedFlatIndex: row_groups = %d does not divide the %d ranks" % (self.row_groups, self.world_size))
        if self.row_groups > MAX_SHARDS:
            raise ValueError("ShardedFlatIndex: %d row shards, but trx_merge_topk_device merges at most %d lists "
                             "(include/trx_knn.h); shard over the GPUs of one node" % (self.row_groups, MAX_SHARDS))
        self.query_groups = self.world_size // self.row_groups
        self.row_rank, self.query_rank = self.rank % self.row_groups, self.rank // self.row_groups
        self.col_group = group
        if self.query_groups > 1:
            if group is not None:
                raise ValueError("ShardedFlatIndex: a rows x queries grid is laid over the default process group")
            # every rank creates every column's group (new_group is collective); the column of rank r is ranks [j Gr, (j + 1) Gr)
            for j in range(self.query_groups):
                g = dist.new_group(list(range(j * self.row_groups, (j + 1) * self.row_groups)))
                if j == self.query_rank:
                    self.col_group = g
        if local_index is None:
            from . import faiss_compat
            local_index = faiss_compat.IndexFlat(d, metric)
        if merge is None:
            from . import faiss_compat
            merge = faiss_compat.merge_topk
        self.local = local_index
        self._merge = merge
        # FAISS' order among exact inner-product ties (include/trx_knn.h, TRX_TIES_FAISS) is a property of the WHOLE index:
        # the shards answer with their canonical top 2k, the merge keeps the canonical top 2k of the union, and the rule is
        # applied once, after the merge -- as one FAISS index over all rows would answer.  The local index HANDS ITS RULE
        # OVER: it answers by id order from here on (also when used directly), until `release_local()` gives the rule back.
        self.tie_rule = getattr(local_index, "tie_rule", "id")
        if self.tie_rule == "faiss":
            import inspect
            try:
                params = inspect.signature(merge).parameters
                takes = "faiss_ties_k" in params or any(p.kind == p.VAR_KEYWORD for p in params.values())
            except (TypeError, ValueError):
                takes = True
            if not takes:      # found here, not as a TypeError in the middle of a collective
                raise TypeError("ShardedFlatIndex: tie_rule='faiss' needs a merge callable that takes faiss_ties_k=")
            local_index.set_tie_rule("id")
        # a one-rank group skips the exchange (nothing to exchange); exchange_always runs the collectives and the merge
        # anyway, which is how a one-GPU box drives the RCCL branch of this file (tests/test_knn_gpu.py)
        self.exchange_always = bool(exchange_always)
        # stream_ordered (default): the exchange is enqueued behind the local search without a host wait, and ONE extra collective -- a
        # one-word all-reduce after the wait -- makes the ranks agree on whether any of them had to re-do queries late (then the
        # exchange repeats).  False: the local search is finished first (one host wait), then the exchange runs on final lists:
        # no agreement collective, no repeat, at the price of the host's wake-up latency between the scan and the all-to-all.  Which
        # is cheaper at 8 ranks has never been measured (no multi-GPU node): bench.py --finish-first is the A/B for the first run.
        self.stream_ordered = bool(stream_ordered)
        self.late_fallbacks = 0
        self.offset = 0      # global id of this shard's first row
        self.ntotal = 0      # rows over all shards

    def release_local(self):
        """give the local index its own tie rule back (the constructor took it over); the wrapper must not be used afterwards"""
        if self.tie_rule == "faiss":
            self.local.set_tie_rule("faiss")
        return self.local

    def shard_rows(self, n_total: int) -> Tuple[int, int]:
        """the rows of an n_total-row corpus this rank holds: row shard `row_rank` of `row_groups`"""
        return shard_bounds(n_total, self.row_groups, self.row_rank)

    def add_shard(self, x_local, offset: int, ntotal: int):
        """Add this rank's rows (already partitioned by the caller: `shard_rows`, or shard_bounds for pure row sharding)."""
        self.local.add(x_local)
        self.offset, self.ntotal = int(offset), int(ntotal)

    def search(self, x, k: int):
        """x replicated on every rank (torch tensor on this rank's device).  Returns (D, I) with global ids, identical on
        every rank.

        Two collectives, both a G-th of what an all-gather of every rank's full list would move (the all-to-all
        variant): rank r receives, from every rank of its column, the lists of ITS share of the column's queries (one
        all-to-all of nq*k*16 bytes per rank in total), merges those queries, and an all-gather of the merged (D, I) slices
        (nq*k*12 bytes per rank) replicates the result.  With the HIP index the local search is stream-ordered
        (trx_index_search_device_begin): the exchange and the merge are enqueued behind it without a host round trip, and ONE
        wait at the end reads the certificate counts back; should that wait have had to re-do queries after the exchange
        (stats.late_fallback on any rank -- agreed on by a one-word all-reduce), the exchange is simply repeated on the final
        lists."""
        import torch
        k = int(k)
        exchange = self.world_size > 1 or self.exchange_always
        ties = self.tie_rule == "faiss" and self.metric == 0
        if ties and 2 * k > MAX_K:
            raise AssertionError("tie_rule='faiss' searches for the canonical top 2k: k must be in [1, %d]" % (MAX_K // 2))
        if ties and not exchange:       # one shard, nothing to merge: the local index applies the rule itself
            self.local.set_tie_rule("faiss")
            try:
                D, I_loc = self.local.search(x, k)
            finally:
                self.local.set_tie_rule("id")
            return D, torch.where(I_loc >= 0, I_loc + self.offset, I_loc)
        k_out, k = k, (2 * k if ties else k)
        nq = x.shape[0]
        qlo, qhi = shard_bounds(nq, self.query_groups, self.query_rank)      # this column's slice of the queries (all of them: Gq = 1)
        xs = x if self.query_groups == 1 else x[qlo:qhi].contiguous()
        begin = getattr(self.local, "search_s64_begin", None) if (exchange and self.stream_ordered) else None
        D, I_loc, S = begin(xs, k) if begin is not None else self.local.search_s64(xs, k)
        if not exchange:
            return D, torch.where(I_loc >= 0, I_loc + self.offset, I_loc)
        ties_k = k_out if ties else None
        out = self._exchange_and_merge(S, I_loc, k, ties_k, nq)
        if begin is not None:
            import torch.distributed as dist
            mine_late = 1 if self.local.search_finish() else 0
            self.late_fallbacks += mine_late          # searches of THIS rank that had to repeat the exchange (tests read it)
            late = torch.tensor([mine_late], dtype=torch.int32)
            if dist.get_backend(self.group) != "gloo":
                late = late.to(S.device)
            dist.all_reduce(late, op=dist.ReduceOp.MAX, group=self.group)
            if int(late.item()):
                out = self._exchange_and_merge(S, I_loc, k, ties_k, nq)
        return out

    def _exchange_and_merge(self, S, I_loc, k, ties_k, nq_all):
        import torch
        import torch.distributed as dist
        G, r, nq = self.row_groups, self.row_rank, S.shape[0]      # the column: Gr ranks, this column's nq queries
        I = torch.where(I_loc >= 0, I_loc + self.offset, I_loc)
        # query-major pack [nq, 2, k] (fp64 score bits, global id): the rows of destination j are contiguous
        pack = torch.stack([S.contiguous().view(torch.int64), I.contiguous()], dim=1)
        bounds = [shard_bounds(nq, G, j) for j in range(G)]
        mine = bounds[r][1] - bounds[r][0]
        gloo = dist.get_backend(self.col_group) == "gloo"       # test path: gloo moves host tensors
        src = pack.cpu() if gloo else pack
        got = torch.empty((G * mine, 2, k), dtype=torch.int64, device=src.device)
        dist.all_to_all_single(got, src, output_split_sizes=[mine] * G, input_split_sizes=[hi - lo for lo, hi in bounds],
                               group=self.col_group)
        got = got.to(pack.device).view(G, mine, 2, k)
        if ties_k:
            Dm, Im = self._merge(self.metric, got[:, :, 0].contiguous().view(torch.float64), got[:, :, 1].contiguous(), faiss_ties_k=ties_k)
            k = ties_k
        else:
            Dm, Im = self._merge(self.metric, got[:, :, 0].contiguous().view(torch.float64), got[:, :, 1].contiguous())
        # where every rank's merged queries lie among ALL queries: rank j Gr + i has share i of column j's slice
        all_bounds = []
        for j in range(self.query_groups):
            cl, ch = shard_bounds(nq_all, self.query_groups, j)
            all_bounds += [(cl + lo, cl + hi) for lo, hi in (shard_bounds(ch - cl, G, i) for i in range(G))]
        return _gather_query_slices(Dm, Im, all_bounds, mine, k, self.group)


def _gather_query_slices(Dm, Im, bounds, mine, k, group):
    """every rank holds the final (D, I) of ITS slice of the queries (bounds[rank]) -> all of them on every rank: slices are
    at most one query apart in length; pad to the longest, one all-gather"""
    import torch
    import torch.distributed as dist
    G = len(bounds)
    gloo = dist.get_backend(group) == "gloo"       # test path: gloo moves host tensors
    per = max(hi - lo for lo, hi in bounds)
    buf = torch.zeros((per, 2, k), dtype=torch.int64, device=Dm.device)
    buf[:mine, 0] = Dm.contiguous().view(torch.int32).long()      # the float bits, widened: one tensor, one collective
    buf[:mine, 1] = Im
    if gloo:
        host = buf.cpu()
        parts = [torch.empty_like(host) for _ in range(G)]
        dist.all_gather(parts, host, group=group)
        allb = torch.stack(parts).to(Dm.device)
    else:
        allb = torch.empty((G,) + tuple(buf.shape), dtype=torch.int64, device=Dm.device)
        dist.all_gather_into_tensor(allb, buf, group=group)
    rows = torch.cat([allb[j, :hi - lo] for j, (lo, hi) in enumerate(bounds)])
    return rows[:, 0].to(torch.int32).view(torch.float32), rows[:, 1].contiguous()


class ReplicatedFlatIndex:
    """The other way to use G GPUs, FAISS' own default (`faiss.index_cpu_to_all_gpus` builds an IndexReplicas unless
    `co.shard` is set): every rank holds ALL rows in its own flat index and searches rank r's 1/G of the queries
    (`shard_bounds(nq, G, r)`); one all-gather of the (D, I) slices replicates the result.  No exchange of candidates, no
    merge, and the per-(query tile, corpus split) warm-up of the scan is paid on a G-th of the query tiles: at 8 ranks one
    rank's share of the headline search is 9.95 ms against 11.3 for the row shard.  For a corpus that
    fits one GPU; `ShardedFlatIndex` is the form for one that does not, and the one `north_star` names.
    Same results as one flat index, on every rank: queries are independent."""

    def __init__(self, d: int, metric: int, group=None, local_index=None):
        import torch.distributed as dist
        self.d, self.metric, self.group = int(d), int(metric), group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if local_index is None:
            from . import faiss_compat
            local_index = faiss_compat.IndexFlat(d, metric)
        self.local = local_index

    @property
    def ntotal(self):
        return self.local.ntotal

    def add(self, x):
        """all rows, on every rank"""
        self.local.add(x)

    def search(self, x, k: int):
        """x replicated on every rank (torch tensor on this rank's device) -> (D, I), identical on every rank"""
        nq = x.shape[0]
        if self.world_size == 1:
            return self.local.search(x, k)
        bounds = [shard_bounds(nq, self.world_size, j) for j in range(self.world_size)]
        lo, hi = bounds[self.rank]
        D, I = self.local.search(x[lo:hi].contiguous(), k)
        return _gather_query_slices(D, I, bounds, hi - lo, k, self.group)
