"""Brute-force Tanimoto retrieval on the GPU: the reference's retrieve/retrieve.py.

retrieve.py:34-40 scores ONE test reaction against every train reaction with RDKit's TanimotoSimilarity in a pool of
64 processes, :55-62 keeps the 100 best (`rank`, `similarity`) and dumps {row: {...}} to test_nn.json.  Here the
train fingerprints -- the dense count arrays retrieve_faiss.py:24-27 already builds from the same RDKit difference
fingerprints -- are packed once into HBM (one byte per count magnitude, 64-row blocks) and libtrxtani.so
(include/trx_tanimoto.h) scores 64 queries per pass over them with v_sad_u8; a bound from per-block maxima leaves a short
list per query for which exact integer keys are formed; the similarities come back as the same doubles (integer
numerator / integer denominator, one IEEE division) and the ranking is exact.

    index = TanimotoIndex(2048)                      # cuda:0
    index.add(train_fps)                             # [N, 2048] int (numpy or torch, host or device)
    sim, rank = index.search(test_fps, k=100)        # float64 [Q, 100], int64 [Q, 100]
    result = retrieve(test_fps, train_fps)           # {i: {'rank': [...], 'similarity': [...]}} as retrieve.py:60-63

No CPU fallback: without the HIP library or a GPU the calls raise.  Tokenisation of SMILES into fingerprints is RDKit's
job (absent here) and stays with the reference's own functions.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_CSRC, "libtrxtani.so")

# every symbol include/trx_tanimoto.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = ["trx_tanimoto_packed_bytes", "trx_tanimoto_pack", "trx_tanimoto_scores", "trx_tanimoto_filter", "trx_tanimoto_last_error"]
I64, I32, I8 = 0, 1, 2
QUERY_GROUP, KEY_ID_BITS = 16, 27
MAX_SUM = 32768          # sum |count| per fingerprint must stay below this (exactness of the key order, see the header)


class TrxTanimotoError(RuntimeError):
    pass


_lib = None


def build(force=False):
    src = os.path.join(_CSRC, "tanimoto.hip")
    hdr = os.path.join(_HERE, "..", "include", "trx_tanimoto.h")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _CSRC, "libtrxtani.so"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            try:
                build()
            except Exception as e:
                raise TrxTanimotoError("libtrxtani.so is missing and could not be built: %s" % e)
        L = ctypes.CDLL(_SO)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.trx_tanimoto_packed_bytes.restype = i64
        L.trx_tanimoto_packed_bytes.argtypes = [i64, i32]
        L.trx_tanimoto_pack.argtypes = [vp, i32, i64, i32, i64, i64, vp, vp, vp, vp]
        L.trx_tanimoto_scores.argtypes = [vp, vp, i64, i32, vp, vp, i32, vp, i64, vp, vp]
        L.trx_tanimoto_filter.argtypes = [vp, i64, vp, vp, vp, i32, i64, vp, i64, vp, vp, vp]
        L.trx_tanimoto_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise TrxTanimotoError(lib().trx_tanimoto_last_error().decode() or "libtrxtani error %d" % rc)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


_DT = {torch.int64: I64, torch.int32: I32, torch.int8: I8}


class TanimotoIndex:
    """flat store of count fingerprints in HBM; `search` = retrieve.py:34-40 + :59-62 for a batch of queries"""

    def __init__(self, d=2048, device=0):
        if d <= 0 or d % 4:
            raise TrxTanimotoError("the fingerprint length must be a positive multiple of 4 (got %d)" % d)
        if not torch.cuda.is_available():
            raise TrxTanimotoError("TanimotoIndex needs a GPU (there is no CPU implementation behind it)")
        lib()
        self.d, self.dev = d, torch.device("cuda", device)
        self.ntotal, self.cap = 0, 0
        self.packed = None          # uint8 [cap / 64 * d * 64] in the layout of the header
        self._tail = None           # the rows of a partially filled last block, as they were handed in
        self.row_sum = None         # int32 [cap]

    def _as_device_ints(self, x):
        t = torch.as_tensor(np.ascontiguousarray(x)) if not torch.is_tensor(x) else x
        if t.dim() != 2 or t.shape[1] != self.d:
            raise TrxTanimotoError("expected [n, %d] counts, got %s" % (self.d, tuple(t.shape)))
        if t.dtype == torch.uint8 or t.dtype == torch.bool or t.dtype == torch.int16:
            t = t.to(torch.int32)
        if t.dtype not in _DT:
            raise TrxTanimotoError("fingerprints must be integer counts (got %s)" % t.dtype)
        return t.to(self.dev).contiguous()

    def _reserve(self, n):
        if n <= self.cap:
            return
        cap = max(((n + 63) // 64) * 64, 2 * self.cap)
        packed = torch.zeros(cap * self.d, dtype=torch.uint8, device=self.dev)
        row_sum = torch.zeros(cap, dtype=torch.int32, device=self.dev)
        if self.ntotal:
            used = ((self.ntotal + 63) // 64) * 64
            packed[:used * self.d] = self.packed[:used * self.d]
            row_sum[:self.ntotal] = self.row_sum[:self.ntotal]
        self.packed, self.row_sum, self.cap = packed, row_sum, cap

    def add(self, fps):
        """append [n, d] integer counts (signed: magnitudes are what RDKit's similarity uses)"""
        t = self._as_device_ints(fps)
        if t.shape[0] == 0:
            return
        if self.ntotal + t.shape[0] >= (1 << KEY_ID_BITS):
            raise TrxTanimotoError("at most 2^27 - 1 rows")
        first = self.ntotal
        if self._tail is not None:
            # the last block of 64 rows is partially filled: it is packed again, its old rows in front of the new ones
            first = self.ntotal - self._tail.shape[0]
            t = torch.cat([self._tail.to(t.dtype), t])
        n = t.shape[0]
        self._reserve(first + n)
        self.row_sum[first:self.ntotal] = 0
        flags = torch.zeros(1, dtype=torch.int32, device=self.dev)
        _check(lib().trx_tanimoto_pack(t.data_ptr(), _DT[t.dtype], n, self.d, t.stride(0), first, self.packed.data_ptr(),
                                       self.row_sum.data_ptr(), flags.data_ptr(), _stream(self.dev)))
        bad = None
        if int(flags.item()) & 1:
            bad = "a count magnitude above 255 does not fit the byte storage"
        elif int(self.row_sum[first:first + n].max().item()) >= MAX_SUM:
            bad = "sum |count| of a fingerprint must be < %d" % MAX_SUM
        if bad:        # leave the index as it was: the rows that were there are packed back
            self.row_sum[first:first + n] = 0
            if self._tail is not None:
                keep = self._tail
                _check(lib().trx_tanimoto_pack(keep.data_ptr(), _DT[keep.dtype], keep.shape[0], self.d, keep.stride(0), first,
                                               self.packed.data_ptr(), self.row_sum.data_ptr(), flags.data_ptr(), _stream(self.dev)))
            raise TrxTanimotoError(bad)
        self.ntotal = first + n
        rem = self.ntotal % 64
        self._tail = t[n - rem:].clone() if rem else None

    def _pack_queries(self, q):
        mag = q.abs()
        if int(mag.max().item()) > 255:
            raise TrxTanimotoError("a count magnitude above 255 does not fit the byte storage")
        q_sum = mag.sum(dim=1, dtype=torch.int64)
        if int(q_sum.max().item()) >= MAX_SUM:
            raise TrxTanimotoError("sum |count| of a fingerprint must be < %d" % MAX_SUM)
        nq = q.shape[0]
        nq_pad = (nq + QUERY_GROUP - 1) // QUERY_GROUP * QUERY_GROUP
        b = torch.zeros((nq_pad, self.d), dtype=torch.uint8, device=self.dev)
        b[:nq] = mag.to(torch.uint8)
        q_t = b.view(torch.int32).t().contiguous()                  # [d / 4, nq_pad] dwords, dword j of query q at [j, q]
        s = torch.zeros(nq_pad, dtype=torch.int32, device=self.dev)
        s[:nq] = q_sum.to(torch.int32)
        return q_t, s

    def search(self, queries, k=100, batch=128):
        """-> (similarity float64 [Q, k'], rank int64 [Q, k']) on the device, k' = min(k, ntotal); best first, equal
        similarities ordered by descending row number"""
        keys, a, den = self.search_keys(queries, k, batch)
        a, den = a.to(torch.float64), den.to(torch.float64)
        return torch.where(den.abs() < 1e-6, torch.zeros_like(a), a / den), keys & ((1 << KEY_ID_BITS) - 1)

    def search_keys(self, queries, k=100, batch=128):
        """the k' best of every query as (key, and, den) int64 [Q, k']: key = the exact order key of include/trx_tanimoto.h
        (row number in the low 27 bits), similarity = and / den.  What a row-sharded search exchanges and merges."""
        q = self._as_device_ints(queries)
        nq, n = q.shape[0], self.ntotal
        kk = min(k, n)
        keys_out = torch.zeros((nq, kk), dtype=torch.int64, device=self.dev)
        a_out = torch.zeros((nq, kk), dtype=torch.int64, device=self.dev)
        den_out = torch.zeros((nq, kk), dtype=torch.int64, device=self.dev)
        if nq == 0 or kk == 0:
            return keys_out, a_out, den_out
        def run_batch(lo, all_keys, rows=batch):
            """one batch of queries -> its rows of the outputs; returns a device flag "some query had more rows above its
            bound than the candidate list holds" (None when every key was formed anyway).  all_keys: form every key of every
            row (the answer to that flag)"""
            qb = q[lo:lo + rows]
            m = qb.shape[0]
            q_t, q_sum = self._pack_queries(qb)
            both = torch.empty((m, n), dtype=torch.int16, device=self.dev)       # sums of minima (< 32768)
            nblocks = (n + 63) // 64
            cap = max(4 * kk, 1024)
            shortlist = nblocks >= kk and n > 4 * cap and not all_keys      # otherwise every key of a row is formed and selected from directly
            bmax = torch.empty((m, nblocks), dtype=torch.float32, device=self.dev) if shortlist else None
            st = _stream(self.dev)
            _check(lib().trx_tanimoto_scores(self.packed.data_ptr(), self.row_sum.data_ptr(), n, self.d, q_t.data_ptr(), q_sum.data_ptr(),
                                             m, both.data_ptr(), n, bmax.data_ptr() if shortlist else None, st))

            def keys_of(thr, width):
                cand = torch.full((m, width), -1, dtype=torch.int64, device=self.dev)
                counts = torch.zeros(m, dtype=torch.int32, device=self.dev)
                _check(lib().trx_tanimoto_filter(both.data_ptr(), n, self.row_sum.data_ptr(), q_sum.data_ptr(), None, m, n,
                                                 None if thr is None else thr.data_ptr(), width, cand.data_ptr(), counts.data_ptr(), st))
                return cand, counts
            if shortlist:
                # the k-th best block maximum bounds the k-th best similarity from below (header): only rows whose approximate
                # similarity reaches it (less the rounding margin) can be among the k best; exact keys are formed for those only
                thr = (torch.topk(bmax, kk, dim=1, largest=True, sorted=True).values[:, -1] * (1.0 - 2.0 ** -19)).contiguous()
                cand, counts = keys_of(thr, cap)
                over_any = (counts > cap).any()
            else:
                cand, over_any = keys_of(None, n)[0], None
            top = torch.topk(cand, kk, dim=1, largest=True, sorted=True).values
            r = top & ((1 << KEY_ID_BITS) - 1)
            a = both.gather(1, r).to(torch.int64)
            keys_out[lo:lo + m] = top
            a_out[lo:lo + m] = a
            den_out[lo:lo + m] = self.row_sum[:n].to(torch.int64)[r] + q_sum[:m].to(torch.int64)[:, None] - a
            return over_any

        # ONE host read per call, after every batch's outputs are enqueued (the device never waits for the host to decide): a
        # batch in which some query had more rows above its bound than the list holds -- heavily tied data -- is done again
        # from all keys
        flags = [(lo, run_batch(lo, False)) for lo in range(0, nq, batch)]
        flags = [(lo, f) for lo, f in flags if f is not None]
        if flags:
            for (lo, _), again in zip(flags, torch.stack([f for _, f in flags]).tolist()):
                if again:
                    # every key of every row: an [m, n] int64 matrix plus the top-k's workspace -- taken a few queries at a time
                    # so that it stays near 256 MB whatever the shard size (a 10M-row shard: 3 queries per pass)
                    sub = max(1, min(batch, (1 << 25) // max(n, 1)))
                    for l2 in range(lo, min(lo + batch, nq), sub):
                        run_batch(l2, True, min(sub, lo + batch - l2))
        return keys_out, a_out, den_out


class ShardedTanimotoIndex:
    """train rows split over the ranks of a process group (one process per GPU, rows [lo, hi) of
    sharded.shard_bounds on rank r), queries replicated.  A search = local search_keys -> row numbers made global ->
    ONE all-gather of [3, Q, k] int64 (key, and, den) per rank over RCCL -> the k largest of the G * k gathered keys.
    Keys are a total order (similarity, then global row number), so the result equals one unsharded index, on every
    rank.  `local_index` is injectable (anything with add / search_keys / ntotal): the gloo test supplies the oracle."""

    def __init__(self, d=2048, group=None, local_index=None, device=None):
        import torch.distributed as dist
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local = local_index if local_index is not None else TanimotoIndex(d, torch.cuda.current_device() if device is None else device)
        self.offset = self.ntotal = 0

    def add_shard(self, fps_local, offset, ntotal):
        if ntotal >= (1 << KEY_ID_BITS):
            raise TrxTanimotoError("at most 2^27 - 1 rows over all shards")
        self.local.add(fps_local)
        self.offset, self.ntotal = int(offset), int(ntotal)

    def search(self, queries, k=100):
        import torch.distributed as dist
        keys, a, den = self.local.search_keys(queries, k)
        keys = keys + self.offset                                  # the row number sits in the low bits: local -> global
        kk = min(k, self.ntotal)
        if keys.shape[1] < kk:                                     # a shard with fewer than k rows: pad with keys that never win
            pad = (keys.shape[0], kk - keys.shape[1])
            keys = torch.cat([keys, torch.full(pad, -1, dtype=torch.int64, device=keys.device)], dim=1)
            a = torch.cat([a, torch.zeros(pad, dtype=torch.int64, device=a.device)], dim=1)
            den = torch.cat([den, torch.zeros(pad, dtype=torch.int64, device=den.device)], dim=1)
        keys, a, den = keys[:, :kk], a[:, :kk], den[:, :kk]
        if self.world_size > 1:
            pack = torch.stack([keys, a, den]).contiguous()
            if dist.get_backend(self.group) == "gloo":             # test path (gloo gathers host tensors)
                parts = [torch.empty_like(pack.cpu()) for _ in range(self.world_size)]
                dist.all_gather(parts, pack.cpu(), group=self.group)
                out = torch.stack(parts).to(pack.device)
            else:
                out = torch.empty((self.world_size,) + tuple(pack.shape), dtype=torch.int64, device=pack.device)
                dist.all_gather_into_tensor(out, pack, group=self.group)
            allk = out[:, 0].permute(1, 0, 2).reshape(keys.shape[0], -1)          # [Q, G * k]
            top, pos = torch.topk(allk, kk, dim=1, largest=True, sorted=True)
            keys = top
            a = out[:, 1].permute(1, 0, 2).reshape(keys.shape[0], -1).gather(1, pos)
            den = out[:, 2].permute(1, 0, 2).reshape(keys.shape[0], -1).gather(1, pos)
        a, den = a.to(torch.float64), den.to(torch.float64)
        return torch.where(den.abs() < 1e-6, torch.zeros_like(a), a / den), keys & ((1 << KEY_ID_BITS) - 1)


def retrieve(test_fps, train_fps, k=100, limit=None, device=0):
    """the structure retrieve.py:55-66 dumps to test_nn.json: {i: {'rank': [...], 'similarity': [...]}} for the first
    `limit` queries (the script stops after 100)"""
    index = TanimotoIndex(np.asarray(train_fps).shape[1] if not torch.is_tensor(train_fps) else train_fps.shape[1], device)
    index.add(train_fps)
    q = test_fps if limit is None else test_fps[:limit]
    sim, rank = index.search(q, k)
    sim, rank = sim.cpu().numpy(), rank.cpu().numpy()
    return {i: {"rank": rank[i].tolist(), "similarity": sim[i].tolist()} for i in range(len(rank))}


def write_results(path, results):
    """json.dump as retrieve.py:65-66 does (the integer query numbers become JSON strings, as there)"""
    import json
    with open(path, "w") as f:
        json.dump(results, f)


def retrieve_sharded(test_fps, train_fps, k=100, limit=None, local_index=None):
    """`retrieve` with the train rows split over the ranks of the initialised process group (rank r indexes rows
    sharded.shard_bounds(n, G, r); `train_fps` may be a memory-mapped array: a rank reads its rows only): the same
    structure on every rank, equal to what one GPU returns"""
    import torch.distributed as dist
    from .sharded import shard_bounds
    n, d = train_fps.shape
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(n, world, rank)
    index = ShardedTanimotoIndex(d, local_index=local_index)
    index.add_shard(np.asarray(train_fps[lo:hi]), lo, n)
    q = np.asarray(test_fps if limit is None else test_fps[:limit])
    sim, rank_ = index.search(q, k)
    sim, rank_ = sim.cpu().numpy(), rank_.cpu().numpy()
    return {i: {"rank": rank_[i].tolist(), "similarity": sim[i].tolist()} for i in range(len(rank_))}


def main(argv=None):
    """retrieve.py's job on fingerprint arrays that already exist (retrieve_faiss.py caches them as train_fp.pkl):
    python -m textreact_amd.tanimoto --train_fps train_fp.pkl --test_fps test_fp.pkl --output test_nn.json [--limit 100]
    Launched by `python -m torch.distributed.run --nproc-per-node G -m textreact_amd.tanimoto ...` the train rows are split
    over the G GPUs (ShardedTanimotoIndex: one all-gather of keys over RCCL) and rank 0 writes the same file."""
    import argparse
    import os
    import pickle
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--train_fps", required=True, help=".npy or pickle of an [N, d] integer array")
    ap.add_argument("--test_fps", required=True)
    ap.add_argument("--output", default="test_nn.json")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--limit", type=int, default=100, help="number of test rows (retrieve.py stops after 100); -1 = all")
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def load(p, mmap=False):
        if p.endswith(".npy"):
            return np.load(p, mmap_mode="r" if mmap else None)
        with open(p, "rb") as f:
            try:
                return np.asarray(pickle.load(f))
            except Exception:          # retrieve_faiss.py writes train_fp.pkl with np.save (NumPy bytes despite the name)
                f.seek(0)
                return np.load(f)
    limit = None if a.limit < 0 else a.limit
    if world <= 1:
        write_results(a.output, retrieve(load(a.test_fps), load(a.train_fps), k=a.k, limit=limit))
        return 0
    import torch.distributed as dist
    from . import _dist
    _dist.setup()          # set_device, then init_process_group("nccl", device_id=...): textreact_amd/_dist.py
    res = retrieve_sharded(load(a.test_fps), load(a.train_fps, mmap=True), k=a.k, limit=limit)
    if dist.get_rank() == 0:
        write_results(a.output, res)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
