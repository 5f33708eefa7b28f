"""On-the-fly retrieval inside the trainer: BASELINE.json configs[4] ("DDP ... with on-the-fly HIP retrieval each epoch"),
SURVEY.md section 7 step 8 and 8f rank 3.

The reference never retrieves while it trains: it reads three static neighbor files once (main.py:311-323) that an
external Tevatron bi-encoder + FAISS run produced beforehand (README.md:44-47), and its dataset turns a sample's neighbor
ids into encoder input text at __getitem__ time (textreact/dataset.py:58-80, 173-184, 222-236).  Here the same chain runs
on the GPUs at the start of an epoch, nothing through the host or the disk:

  refresh_neighbors   passages row-sharded over the ranks -> dense.encode (BERT [CLS] embeddings through the HIP attention /
                      add+LayerNorm kernels) -> ShardedFlatIndex (libtrxknn.so per rank, one all-gather of (fp64 score, id),
                      HIP merge); every rank encodes 1/G of the queries and ONE all-gather replicates their embeddings
                      -> neighbor ids [N, k], identical on every rank and equal to what one flat index over all
                      passage embeddings returns (tests/test_live_gpu.py checks them against oracle.knn_canonical)
  select_neighbors    dataset.py:58-80 on tensors: ids known to the corpus, the gold passage first (--use_gold_neighbor),
                      text de-duplication (:46-56), --max_num_neighbors, the --random_neighbor_ratio draw, and for the
                      second evaluation loader the gold TEXT removed (:74-76)
  assemble_inputs     what `enc_tokenizer(smiles, text_pair=' (0) t0 (1) t1 (2) t2')` followed by the truncation to
                      --max_length yields (dataset.py:182-183, 147-152), from pre-tokenised pieces: [CLS] query [SEP]
                      marker_0 passage_0 marker_1 passage_1 ... [SEP].  BERT's tokenizer splits on whitespace and
                      punctuation before WordPiece, so the pieces tokenised apart concatenate to the tokens of the whole.
  apply_mlm           dataset.py:82-122: Poisson(3) span masking up to int(len * mlm_ratio) tokens in 100 draws, masked
                      tokens moved to the front with their original positions as position_ids

Tokenizers stay out of scope: the corpus file holds token ids --

    torch.save({"passage_ids": LongTensor [P, Lp] (padded), "passage_len": LongTensor [P],
                "marker_ids": LongTensor [M, m] (the tokens of " (0)", " (1)", ...; M >= num_neighbors),
                "cls_id": int, "sep_id": int, "pad_id": int, "mask_id": int}, path)

and a split's tensor file carries `query_ids` [N, Lq], `query_len` [N] (the reaction / product SMILES tokens alone) and
optionally `gold_passage` [N] (row of the sample's own passage in the corpus, -1 = none: `rxn_id in self.corpus`).
Everything below the two encoders and the index is index arithmetic on device tensors (torch ops); it also runs on CPU
tensors, which is how tests/test_live_cpu.py checks it against neighbors.NeighborStore, the mirror of the reference's
dataset logic that is pinned on outputs of the reference itself.
"""
import torch

from .sharded import ShardedFlatIndex, shard_bounds


class LiveCorpus:
    def __init__(self, path_or_dict, device="cpu"):
        d = torch.load(path_or_dict, map_location="cpu", weights_only=False) if isinstance(path_or_dict, str) else path_or_dict
        self.device = torch.device(device)
        self.passage_ids = d["passage_ids"].long().to(self.device)
        self.passage_len = d["passage_len"].long().to(self.device)
        self.marker_ids = d["marker_ids"].long().to(self.device)
        self.cls_id, self.sep_id, self.pad_id = int(d["cls_id"]), int(d["sep_id"]), int(d["pad_id"])
        self.mask_id = int(d.get("mask_id", -1))
        P, Lp = self.passage_ids.shape
        assert self.passage_len.shape == (P,) and int(self.passage_len.max()) <= Lp
        # passages with the same TEXT (deduplicate_neighbors compares corpus[i] == corpus[j]): same token row
        ar = torch.arange(Lp, device=self.device)[None]
        canon = torch.where(ar < self.passage_len[:, None], self.passage_ids, torch.full_like(self.passage_ids, -1))
        self.group = torch.unique(canon, dim=0, return_inverse=True)[1]

    def __len__(self):
        return self.passage_ids.shape[0]


def with_special_tokens(ids, lens, cls_id, sep_id, pad_id):
    """[n, W] token rows of lengths `lens` -> ([CLS] row [SEP], mask) padded to W + 2"""
    n, W = ids.shape
    out = torch.full((n, W + 2), pad_id, dtype=torch.long, device=ids.device)
    out[:, 0] = cls_id
    ar = torch.arange(W, device=ids.device)[None]
    out[:, 1:W + 1] = torch.where(ar < lens[:, None], ids, torch.full_like(ids, pad_id))
    out.scatter_(1, (lens + 1)[:, None], torch.full((n, 1), sep_id, dtype=torch.long, device=ids.device))
    mask = (torch.arange(W + 2, device=ids.device)[None] < (lens + 2)[:, None]).long()
    width = int(lens.max()) + 2 if n else 2
    return out[:, :width], mask[:, :width]


def all_gather_rows(x, n_total, rank, world, group=None, always=False):
    """rows [lo, hi) of an [n_total, H] matrix per rank (shard_bounds) -> the whole matrix on every rank: one collective
    (always=True: also in a one-rank group, which is how a one-GPU box drives the RCCL branch)"""
    if world == 1 and not always:
        return x
    import torch.distributed as dist
    per = -(-n_total // world)
    buf = torch.zeros((per, x.shape[1]), dtype=x.dtype, device=x.device)
    buf[:x.shape[0]] = x
    if dist.get_backend(group) == "gloo":      # test path (gloo gathers host tensors; the rows travel as bytes)
        host = buf.cpu().view(torch.uint8)
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        parts = [p.view(buf.dtype).to(x.device) for p in parts]
    else:
        out = torch.empty((world * per, x.shape[1]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, buf, group=group)
        parts = list(out.view(world, per, -1))
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        rows.append(parts[r][:hi - lo])
    return torch.cat(rows)


class LiveRetriever:
    """the passage side (embeddings of this rank's rows in a flat index, rebuilt by `refresh_index`) and the query side
    (`neighbors`) of the on-the-fly retrieval.  local_index / merge: injectable like ShardedFlatIndex's (CPU tests)."""

    def __init__(self, corpus, rank=0, world=1, group=None, batch_size=256, local_index=None, merge=None, autocast=True,
                 timing=False):
        self.corpus, self.rank, self.world, self.group = corpus, rank, world, group
        self.batch_size, self.autocast = batch_size, autocast
        self._local_index, self._merge = local_index, merge
        self.index, self.emb_p = None, None
        # timing=True: every stage is bracketed by a device synchronisation and its wall time lands in `ms`
        # (bench_predictor.py --live, BASELINE.md C4 "retrieval refresh time per epoch"); off, nothing waits
        self.timing, self.ms = timing, {}

    def _lap(self, name, t0):
        if not self.timing:
            return None
        import time
        dev = self.corpus.device
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        if t0 is not None:
            self.ms[name] = self.ms.get(name, 0.0) + (t1 - t0) * 1e3
        return t1

    def refresh_index(self, p_encoder):
        """encode this rank's rows [lo, hi) of the corpus with the encoder's CURRENT weights and index them"""
        from . import dense
        c = self.corpus
        lo, hi = shard_bounds(len(c), self.world, self.rank)
        t = self._lap(None, None)
        pid, pmask = with_special_tokens(c.passage_ids[lo:hi], c.passage_len[lo:hi], c.cls_id, c.sep_id, c.pad_id)
        self.emb_p = dense.encode(p_encoder, pid, pmask, batch_size=self.batch_size, autocast=self.autocast,
                                  lengths=c.passage_len[lo:hi] + 2)
        t = self._lap("encode_passages", t)
        li = self._local_index() if callable(self._local_index) else self._local_index
        if li is None:      # the HIP flat index, on the device the embeddings live on (not LOCAL_RANK: a rehearsal shares one GPU)
            from . import faiss_compat
            li = faiss_compat.IndexFlat(self.emb_p.shape[1], faiss_compat.METRIC_INNER_PRODUCT, device=self.emb_p.device.index or 0)
        self.index = ShardedFlatIndex(self.emb_p.shape[1], 0, group=self.group, local_index=li, merge=self._merge)
        self.index.add_shard(self.emb_p, lo, len(c))
        self._lap("index_add", t)

    def embed_queries(self, q_encoder, query_ids, query_len):
        """every rank encodes 1/G of the queries; ONE all-gather replicates the embeddings [N, H]"""
        from . import dense
        c, N, dev = self.corpus, query_ids.shape[0], self.emb_p.device
        lo, hi = shard_bounds(N, self.world, self.rank)
        t = self._lap(None, None)
        qlen = query_len[lo:hi].to(dev)
        qid, qmask = with_special_tokens(query_ids[lo:hi].to(dev), qlen, c.cls_id, c.sep_id, c.pad_id)
        e = dense.encode(q_encoder, qid, qmask, batch_size=self.batch_size, autocast=self.autocast, lengths=qlen + 2)
        t = self._lap("encode_queries", t)
        out = all_gather_rows(e, N, self.rank, self.world, self.group)
        self._lap("gather_queries", t)
        return out

    def neighbors(self, q_encoder, query_ids, query_len, k):
        """[N, k] rows of the corpus, best first (-1 = fewer than k passages), identical on every rank"""
        emb_q = self.embed_queries(q_encoder, query_ids, query_len)
        t = self._lap(None, None)
        _, nn = self.index.search(emb_q, k)
        self._lap("search", t)
        return nn


def refresh_neighbors(q_encoder, p_encoder, corpus, query_ids, query_len, k, rank=0, world=1, group=None, batch_size=256,
                      local_index=None, merge=None, autocast=True, return_embeddings=False, timings=None):
    """one-call form: index the passages, search every query -> neighbor ids [N, k], identical on every rank.
    timings: a dict that receives the wall milliseconds of every stage (each bracketed by a device synchronisation)"""
    r = LiveRetriever(corpus, rank, world, group, batch_size, local_index, merge, autocast, timing=timings is not None)
    r.refresh_index(p_encoder)
    emb_q = r.embed_queries(q_encoder, query_ids, query_len)
    t = r._lap(None, None)
    _, nn = r.index.search(emb_q, k)
    r._lap("search", t)
    if timings is not None:
        timings.update(r.ms)
        stats = getattr(r.index.local, "last_stats", None)
        if stats is not None:      # queries the fast path could not certify (re-done by the exact fp64 scan, this rank's shard)
            st = stats()
            timings["uncertified_queries"] = st["n_uncertified"]
            timings["rescored_queries"] = st.get("n_rescored", 0)
            timings["rescanned_queries"] = st.get("n_rescanned", 0)
    return (nn, emb_q, r.emb_p) if return_embeddings else nn


def select_neighbors(nn, gold, corpus, train, use_gold_neighbor=False, max_num_neighbors=10, num_neighbors=3,
                     random_neighbor_ratio=0.0, skip_gold_neighbor=False, generator=None):
    """dataset.py:58-80 for all samples at once: nn [N, k] retrieved passage rows (-1 = none), gold [N] the sample's own
    passage row (-1 = not in the corpus) -> [N, num_neighbors] passage rows, -1 padded, in the order their texts enter
    the encoder input."""
    dev = nn.device
    N = nn.shape[0]
    gold = gold.to(dev) if gold is not None else torch.full((N,), -1, dtype=torch.long, device=dev)
    ids = nn.clone()
    keep = ids >= 0
    grp = corpus.group.to(dev)
    if train and use_gold_neighbor:                      # :62-66: gold id removed where it was retrieved, then put first
        keep &= ~((ids == gold[:, None]) & (gold[:, None] >= 0))
        ids = torch.cat([gold[:, None], ids], dim=1)
        keep = torch.cat([(gold >= 0)[:, None], keep], dim=1)
    if (not train) and skip_gold_neighbor:               # :74-76: every neighbour whose TEXT is the gold text goes
        gg = torch.where(gold >= 0, grp[gold.clamp(min=0)], torch.full_like(gold, -2))
        keep &= grp[ids.clamp(min=0)] != gg[:, None]
    # :46-56 first id of every distinct text, order kept
    g = torch.where(keep, grp[ids.clamp(min=0)], -1 - torch.arange(ids.shape[1], device=dev)[None].expand_as(ids))
    same = (g[:, :, None] == g[:, None, :]) & torch.tril(torch.ones(ids.shape[1], ids.shape[1], dtype=torch.bool, device=dev), -1)[None]
    keep &= ~same.any(dim=2)
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)        # kept ids to the front, order kept
    ids = torch.gather(ids, 1, order)
    cnt = keep.sum(dim=1)
    width = ids.shape[1]
    ar = torch.arange(width, device=dev)[None]
    if train:
        cnt = cnt.clamp(max=max_num_neighbors)                                # :67 neighbors_ids[:max_num_neighbors]
        # :69-72 one draw per sample: a random sample (order included) of the kept texts, or the first num_neighbors
        draw = torch.rand(N, generator=generator, device="cpu").to(dev) < random_neighbor_ratio
        keys = torch.rand(ids.shape, generator=generator, device="cpu").to(dev)
        keys = torch.where(ar < cnt[:, None], keys, torch.full_like(keys, 2.0))
        shuffled = torch.gather(ids, 1, torch.argsort(keys, dim=1))
        ids = torch.where(draw[:, None], shuffled, ids)
    take = cnt.clamp(max=num_neighbors)
    out = torch.where(ar < take[:, None], ids, torch.full_like(ids, -1))[:, :num_neighbors]
    if out.shape[1] < num_neighbors:
        out = torch.cat([out, torch.full((N, num_neighbors - out.shape[1]), -1, dtype=torch.long, device=dev)], dim=1)
    return out


def assemble_inputs(query_ids, query_len, sel, corpus, max_length, with_neighbors=True):
    """[CLS] query [SEP] marker_0 passage_0 marker_1 passage_1 ... [SEP] cut to max_length (dataset.py:182-183, 147-152;
    with_neighbors=False is --num_neighbors 0: [CLS] query [SEP]).  -> input_ids [N, L], attention_mask [N, L], lengths"""
    dev = sel.device
    query_ids, query_len = query_ids.to(dev), query_len.to(dev)
    N = query_ids.shape[0]
    M = sel.shape[1] if with_neighbors else 0
    assert corpus.marker_ids.shape[0] >= M, "the corpus file holds %d neighbour markers, --num_neighbors is %d" % (corpus.marker_ids.shape[0], M)
    mlen = (corpus.marker_ids >= 0).sum(dim=1)                                  # marker rows may be padded with -1
    plen = torch.where(sel >= 0, corpus.passage_len.to(dev)[sel.clamp(min=0)], torch.zeros_like(sel)) if M else None
    total = 2 + query_len
    if with_neighbors:
        total = total + 1
        for j in range(M):
            total = total + torch.where(sel[:, j] >= 0, mlen[j] + plen[:, j], torch.zeros_like(query_len))
    L = int(min(int(total.max()), max_length)) if N else 0
    out = torch.full((N, L), corpus.pad_id, dtype=torch.long, device=dev)
    rows = torch.arange(N, device=dev)

    def put(tokens, lens, off):
        W = tokens.shape[1]
        pos = off[:, None] + torch.arange(W, device=dev)[None]
        m = (torch.arange(W, device=dev)[None] < lens[:, None]) & (pos < L)
        out[rows[:, None].expand_as(pos)[m], pos[m]] = tokens[m]

    one = torch.ones(N, dtype=torch.long, device=dev)
    put(torch.full((N, 1), corpus.cls_id, dtype=torch.long, device=dev), one, torch.zeros(N, dtype=torch.long, device=dev))
    put(query_ids, query_len, one)
    off = 1 + query_len
    put(torch.full((N, 1), corpus.sep_id, dtype=torch.long, device=dev), one, off)
    off = off + 1
    if with_neighbors:
        for j in range(M):
            has = sel[:, j] >= 0
            put(corpus.marker_ids[j].clamp(min=0)[None].expand(N, -1), torch.where(has, mlen[j], torch.zeros_like(off)), off)
            off = off + torch.where(has, mlen[j], torch.zeros_like(off))
            put(corpus.passage_ids.to(dev)[sel[:, j].clamp(min=0)], plen[:, j], off)
            off = off + plen[:, j]
        put(torch.full((N, 1), corpus.sep_id, dtype=torch.long, device=dev), one, off)
    lengths = total.clamp(max=max_length)
    mask = (torch.arange(L, device=dev)[None] < lengths[:, None]).long()
    return out, mask, lengths


def apply_mlm(input_ids, lengths, mlm_ratio, mask_id, generator=None, tries=100):
    """dataset.py:82-122 for all samples at once -> (input_ids with the masked tokens FIRST, position_ids, mlm_labels
    [N, most masked tokens of a sample] padded with -100)"""
    dev = input_ids.device
    N, L = input_ids.shape
    pos = torch.arange(L, device=dev)[None]
    budget = (lengths.float() * mlm_ratio).long()                              # int(len(input_ids) * mlm_ratio)
    masked = torch.zeros((N, L), dtype=torch.bool, device=dev)
    lam = torch.full((N,), 3.0)
    for _ in range(tries):
        kk = torch.poisson(lam, generator=generator).long().to(dev)           # np.random.poisson(lam=3)
        ok = (kk > 0) & (kk <= torch.minimum(torch.full_like(lengths, 10), lengths)) & (kk <= budget) & (lengths - kk > 0)
        u = torch.rand(N, generator=generator).to(dev)
        start = (u * (lengths - kk).clamp(min=1).float()).long()               # random.randrange(input_len - k)
        span = ok[:, None] & (pos >= start[:, None]) & (pos < (start + kk)[:, None])
        masked |= span
        budget = budget - torch.where(ok, kk, torch.zeros_like(kk))
    valid = pos < lengths[:, None]
    labels = torch.where(masked, input_ids, torch.full_like(input_ids, -100))  # span_ids come from the ORIGINAL tokens
    ids = torch.where(masked, torch.full_like(input_ids, mask_id), input_ids)
    # _reorder_masked_sequence: masked tokens first, then the rest, both in their original order; padding stays behind
    rank_key = torch.where(masked, 0, torch.where(valid, 1, 2)).to(torch.int8)
    order = torch.argsort(rank_key, dim=1, stable=True)
    ids = torch.gather(ids, 1, order)
    labels = torch.gather(labels, 1, order)
    position_ids = torch.where(torch.gather(valid, 1, order), order, torch.zeros_like(order))
    trunc = int(masked.sum(dim=1).max()) if N else 0
    return ids, position_ids, labels[:, :trunc].contiguous()
