"""On-the-fly retrieval (textreact_amd/live.py), the host-checkable part: the tensor restatement of the reference's
dataset logic against neighbors.NeighborStore (the mirror of textreact/dataset.py:40-80 that tests/test_neighbors.py pins
on outputs of the reference itself), the input assembly against plain list concatenation, the MLM masking invariants of
dataset.py:82-122, and the 2-rank plumbing (gloo; the oracle stands in for the HIP index and merge, oracle/nn_ref.py for
the HIP ops -- test infrastructure only)."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from textreact_amd import live
from textreact_amd.neighbors import NeighborStore

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLS, SEP, PAD, MASK = 101, 102, 0, 103


def _corpus(P=40, Lp=9, seed=0, dup_every=5):
    g = torch.Generator().manual_seed(seed)
    plen = torch.randint(2, Lp + 1, (P,), generator=g)
    pids = torch.randint(200, 900, (P, Lp), generator=g)
    for i in range(dup_every, P, dup_every):          # passages with the SAME text as an earlier one
        pids[i], plen[i] = pids[i - 3], plen[i - 3]
    pids = torch.where(torch.arange(Lp)[None] < plen[:, None], pids, torch.full_like(pids, 7777))   # junk behind the end
    markers = torch.tensor([[40, 50 + j, 41] for j in range(4)])       # " (j)" -> "(", "j", ")"
    return live.LiveCorpus({"passage_ids": pids, "passage_len": plen, "marker_ids": markers, "cls_id": CLS, "sep_id": SEP,
                            "pad_id": PAD, "mask_id": MASK})


def _store_for(corpus, nn, gold, split, **kw):
    """the same situation as the reference's dataset sees it: ids -> texts, a neighbor dict"""
    texts = {i: " ".join(str(int(t)) for t in corpus.passage_ids[i, :corpus.passage_len[i]]) for i in range(len(corpus))}
    N = nn.shape[0]
    # sample ids: the gold passage's id where the sample has one, otherwise an id outside the corpus
    sample_ids = [int(gold[i]) if int(gold[i]) >= 0 else 10_000 + i for i in range(N)]
    st = NeighborStore(sample_ids, split=split, **kw)
    st.corpus = texts
    st.neighbors = {sample_ids[i]: [int(j) for j in nn[i] if int(j) >= 0] for i in range(N)}
    return st, sample_ids


@pytest.mark.parametrize("split,use_gold,skip_gold", [("train", False, False), ("train", True, False),
                                                      ("val", False, False), ("val", False, True)])
def test_select_neighbors_equals_the_dataset_logic(split, use_gold, skip_gold):
    corpus = _corpus()
    rng = np.random.default_rng(1)
    N, k = 60, 12
    nn = torch.from_numpy(np.stack([rng.permutation(len(corpus))[:k] for _ in range(N)]))
    nn[3, 9:] = -1                                      # fewer than k passages retrieved
    gold = torch.from_numpy(rng.integers(-1, len(corpus), N))
    gold[:20] = nn[:20, rng.integers(0, 9)]             # samples whose own passage was retrieved
    gold[20:30] = -1
    # unique sample ids are needed by the dict-based store: drop duplicated gold ids
    seen = set()
    for i in range(N):
        if int(gold[i]) in seen:
            gold[i] = -1
        seen.add(int(gold[i]))
    st, _ = _store_for(corpus, nn, gold, split, use_gold_neighbor=use_gold, max_num_neighbors=6, num_neighbors=3,
                       random_neighbor_ratio=0.0)
    st.skip_gold_neighbor = skip_gold
    sel = live.select_neighbors(nn, gold, corpus, split == "train", use_gold, 6, 3, 0.0, skip_gold)
    for i in range(N):
        want = st.select_ids(i)[:3]
        got = [int(j) for j in sel[i] if int(j) >= 0]
        assert got == want, (i, got, want)


def test_random_neighbor_branch_draws_a_subset_of_the_kept_texts():
    corpus = _corpus()
    rng = np.random.default_rng(2)
    N, k = 400, 10
    nn = torch.from_numpy(np.stack([rng.permutation(len(corpus))[:k] for _ in range(N)]))
    gold = torch.full((N,), -1, dtype=torch.long)
    first = live.select_neighbors(nn, gold, corpus, True, False, 6, 3, 0.0)
    pool = live.select_neighbors(nn, gold, corpus, True, False, 6, 6, 0.0)
    g = torch.Generator().manual_seed(0)
    drawn = live.select_neighbors(nn, gold, corpus, True, False, 6, 3, 0.5, generator=g)
    changed = 0
    for i in range(N):
        d = [int(j) for j in drawn[i]]
        assert len(set(d)) == 3 and set(d) <= set(int(j) for j in pool[i])        # random.sample of the <= 6 kept ones
        changed += d != [int(j) for j in first[i]]
    assert 0.3 * N < changed < 0.6 * N                   # ~ half the samples take the draw, most draws differ from the top 3


def test_assemble_inputs_is_the_concatenation_the_tokenizer_would_emit():
    corpus = _corpus()
    g = torch.Generator().manual_seed(3)
    N, Lq = 50, 11
    qlen = torch.randint(1, Lq + 1, (N,), generator=g)
    qids = torch.randint(1000, 2000, (N, Lq), generator=g)
    sel = torch.randint(0, len(corpus), (N, 3), generator=g)
    sel[::4, 2] = -1
    sel[::8, 1:] = -1
    sel[5] = -1                                          # no neighbour at all: text_pair = ''
    for max_length in (512, 24):
        ids, mask, lens = live.assemble_inputs(qids, qlen, sel, corpus, max_length)
        for i in range(N):
            want = [CLS] + qids[i, :qlen[i]].tolist() + [SEP]
            for j in range(3):
                if sel[i, j] >= 0:
                    p = int(sel[i, j])
                    want += [40, 50 + j, 41] + corpus.passage_ids[p, :corpus.passage_len[p]].tolist()
            want = (want + [SEP])[:max_length]
            assert ids[i, :lens[i]].tolist() == want and int(lens[i]) == len(want)
            assert ids[i, lens[i]:].eq(PAD).all() and mask[i].tolist() == [1] * len(want) + [0] * (ids.shape[1] - len(want))
    ids, mask, lens = live.assemble_inputs(qids, qlen, sel, corpus, 512, with_neighbors=False)     # --num_neighbors 0
    assert all(ids[i, :lens[i]].tolist() == [CLS] + qids[i, :qlen[i]].tolist() + [SEP] for i in range(N))


def test_mlm_masking_follows_the_reference_rules():
    g = torch.Generator().manual_seed(4)
    N, L = 300, 80
    lens = torch.randint(4, L + 1, (N,), generator=g)
    ids = torch.randint(1000, 2000, (N, L), generator=g)
    ids = torch.where(torch.arange(L)[None] < lens[:, None], ids, torch.zeros_like(ids))
    out, pos, labels = live.apply_mlm(ids, lens, 0.15, MASK, generator=g)
    nmask = out.eq(MASK).sum(dim=1)
    assert (nmask <= (lens.float() * 0.15).long()).all() and nmask.float().mean() > 0.08 * lens.float().mean()
    for i in range(N):
        n, ln = int(nmask[i]), int(lens[i])
        assert out[i, :n].eq(MASK).all() and not out[i, n:ln].eq(MASK).any()           # masked tokens first
        p = pos[i, :ln].tolist()
        assert sorted(p) == list(range(ln)) and p[:n] == sorted(p[:n]) and p[n:] == sorted(p[n:])
        assert labels[i, :n].tolist() == [int(ids[i, j]) for j in p[:n]]                  # labels = the ORIGINAL tokens
        assert labels[i, n:].eq(-100).all()
        assert out[i, n:ln].tolist() == [int(ids[i, j]) for j in p[n:]]                  # the rest in its old order
    assert labels.shape[1] == int(nmask.max())


# ---- two ranks: passages row-sharded, queries encoded 1/2 per rank and all-gathered, sharded search ----------------
class _OracleIndex:
    def __init__(self):
        self.y = None

    def add(self, x):
        self.y = x.float().numpy()

    def search_s64(self, x, k):
        from oracle import flat_knn as oracle
        xq = x.float().numpy()
        D, I = oracle.knn_canonical(0, xq, self.y, k)
        S = np.where(I >= 0, oracle.scores_at(0, xq, self.y, I), -np.finfo(np.float32).max)
        return torch.from_numpy(D), torch.from_numpy(I), torch.from_numpy(S)


def _setup():
    from textreact_amd import dense
    from textreact_amd.predictor.model import Config
    corpus = _corpus(P=57, Lp=12, seed=5)
    torch.manual_seed(0)
    enc = dense.DenseEncoder(Config(vocab_size=2100, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                                    intermediate_size=64, max_position_embeddings=32)).eval()
    g = torch.Generator().manual_seed(6)
    qlen = torch.randint(2, 10, (23,), generator=g)
    qids = torch.randint(1000, 2000, (23, 10), generator=g)
    return corpus, enc, qids, qlen


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import nn_ref
    from test_sharded_gloo import oracle_merge
    nn_ref.install()
    corpus, enc, qids, qlen = _setup()
    nn = live.refresh_neighbors(enc, enc, corpus, qids, qlen, 8, rank, world, local_index=_OracleIndex(), merge=oracle_merge,
                                autocast=False)
    ret[rank] = nn.numpy()
    dist.destroy_process_group()


def test_two_rank_refresh_equals_one_flat_index(reference_ops):
    from oracle import flat_knn as oracle
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    corpus, enc, qids, qlen = _setup()
    nn1, emb_q, emb_p = live.refresh_neighbors(enc, enc, corpus, qids, qlen, 8, local_index=_OracleIndex(),
                                               merge=None, autocast=False, return_embeddings=True)
    _, want = oracle.knn_canonical(0, emb_q.float().numpy(), emb_p.float().numpy(), 8)
    assert np.array_equal(nn1.numpy(), want)
    assert np.array_equal(ret[0], want) and np.array_equal(ret[1], want)


def test_live_retriever_file_must_hold_the_encoder_weights(tmp_path):
    """--live_retriever loaded with strict=False and no check: a file with other key names left a randomly
    initialised retriever behind.  Tevatron's lm_q.* / lm_p.* and a plain encoder.* tree load; anything else stops the run."""
    from textreact_amd import dense
    from textreact_amd.main import load_live_retriever
    from textreact_amd.predictor.model import Config
    cfg = Config(vocab_size=50, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=32, max_position_embeddings=16)
    torch.manual_seed(0)
    a, b = dense.DenseEncoder(cfg), dense.DenseEncoder(cfg)
    tev = {"lm_q." + k[len("encoder."):]: v for k, v in a.state_dict().items()}
    tev.update({"lm_p." + k[len("encoder."):]: v for k, v in b.state_dict().items()})
    tev["lm_q.embeddings.position_ids"] = torch.arange(16)[None]            # a transformers-4.27.3 buffer: tolerated
    torch.save(tev, tmp_path / "tev.pt")
    q, p_ = load_live_retriever(str(tmp_path / "tev.pt"), cfg, "cpu")
    assert all(torch.equal(v, a.state_dict()[k]) for k, v in q.state_dict().items())
    assert all(torch.equal(v, b.state_dict()[k]) for k, v in p_.state_dict().items())
    torch.save({"state_dict": a.state_dict()}, tmp_path / "one.pt")         # one encoder.* tree serves both sides
    q, p_ = load_live_retriever(str(tmp_path / "one.pt"), cfg, "cpu")
    assert all(torch.equal(v, a.state_dict()[k]) for k, v in p_.state_dict().items())
    torch.save({"bert." + k: v for k, v in a.state_dict().items()}, tmp_path / "other.pt")
    with pytest.raises(SystemExit) as e:
        load_live_retriever(str(tmp_path / "other.pt"), cfg, "cpu")
    assert "did not load" in str(e.value)
    part = dict(tev); del part["lm_p.encoder.layer.0.attention.self.query.weight"]
    torch.save(part, tmp_path / "part.pt")
    with pytest.raises(SystemExit):
        load_live_retriever(str(tmp_path / "part.pt"), cfg, "cpu")
