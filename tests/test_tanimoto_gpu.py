"""libtrxtani.so against the oracle (oracle/tanimoto.py): similarities bit-identical doubles, ranks identical including
the order among equal similarities (retrieve/retrieve.py:34-40,55-62)."""
import os

import numpy as np
import pytest
import torch

from oracle import tanimoto as oracle
from test_tanimoto_cpu import fingerprints

pytestmark = pytest.mark.gpu


def _check(corpus, queries, k):
    from textreact_amd.tanimoto import TanimotoIndex
    idx = TanimotoIndex(corpus.shape[1])
    idx.add(corpus)
    sim, rank = idx.search(queries, k)
    want_s, want_r = oracle.search(queries, corpus, k)
    assert np.array_equal(rank.cpu().numpy(), want_r)
    assert np.array_equal(sim.cpu().numpy().view(np.uint64), want_s.view(np.uint64))
    return idx


@pytest.mark.parametrize("n,d,nq,k", [(1000, 2048, 5, 100), (64, 2048, 1, 100), (130, 1024, 17, 10), (4099, 2048, 70, 100),
                                      (777, 64, 33, 7), (5, 8, 3, 100), (20000, 2048, 130, 100), (70001, 1024, 40, 100)])
def test_search_equals_the_oracle(n, d, nq, k):
    rng = np.random.default_rng(n + d + nq)
    corpus = fingerprints(rng, n, d)
    queries = fingerprints(rng, nq, d)
    queries[0] = corpus[n // 2]                    # an exact match
    _check(corpus, queries, k)


def test_dense_fingerprints_with_large_sums_and_crowded_similarities():
    """sums of magnitudes in the ten-thousands and thousands of similarities within 1e-3 of each other: the keys must
    still order them exactly and the fp32 selection bound must not lose a row"""
    rng = np.random.default_rng(8)
    corpus = fingerprints(rng, 20000, 2048, density=0.5, lo=-30, hi=31)
    queries = fingerprints(rng, 20, 2048, density=0.5, lo=-30, hi=31)
    assert 10000 < np.abs(corpus).sum(axis=1).max() < 32768
    _check(corpus, queries, 100)
    near = np.repeat(corpus[:1], 9000, axis=0)                  # 9000 rows that differ from one fingerprint in a few positions
    pos = rng.integers(0, 2048, (9000, 3))
    for i in range(9000):
        near[i, pos[i]] += rng.integers(-2, 3, 3)
    near = np.clip(near, -255, 255)
    _check(near, np.concatenate([corpus[:1], queries[:3]]), 100)


def test_many_equal_similarities_are_ordered_by_descending_row_number():
    rng = np.random.default_rng(3)
    base = fingerprints(rng, 40, 256, density=0.1)
    corpus = base[rng.integers(0, 40, 3000)]       # 3000 rows, only 40 distinct fingerprints
    corpus[100:140] = 0                            # empty fingerprints: similarity 0 by the denominator rule
    queries = np.concatenate([base[:6], np.zeros((1, 256), dtype=np.int64)])
    _check(corpus, queries, 100)


def test_shortlist_selection_with_heavy_ties_falls_back_to_the_full_key_rows():
    """30,000 rows but 40 distinct fingerprints: far more keys sit above the block-maximum bound than the short list
    holds, for some queries; the result is still the oracle's, tie order included"""
    rng = np.random.default_rng(4)
    base = fingerprints(rng, 40, 512, density=0.08)
    corpus = base[rng.integers(0, 40, 30001)]
    queries = np.concatenate([base[:5], fingerprints(rng, 3, 512, density=0.08)])
    _check(corpus, queries, 100)
    _check(corpus, queries, 1)


def test_dtypes_device_inputs_and_the_reference_output_structure():
    from textreact_amd.tanimoto import TanimotoIndex, retrieve
    rng = np.random.default_rng(5)
    corpus = fingerprints(rng, 500, 2048)
    queries = fingerprints(rng, 9, 2048)
    want_s, want_r = oracle.search(queries, corpus, 100)
    for conv in (lambda a: a.astype(np.int32), lambda a: a.astype(np.int8), lambda a: torch.from_numpy(a).cuda()):
        idx = TanimotoIndex(2048)
        idx.add(conv(corpus))
        sim, rank = idx.search(conv(queries), 100)
        assert np.array_equal(rank.cpu().numpy(), want_r) and np.array_equal(sim.cpu().numpy(), want_s)
    res = retrieve(queries, corpus, k=100, limit=4)            # retrieve.py:55-66
    assert sorted(res) == [0, 1, 2, 3] and sorted(res[0]) == ["rank", "similarity"]
    assert res[2]["rank"] == want_r[2].tolist() and res[2]["similarity"] == want_s[2].tolist()


def test_batches_of_any_size_append_and_bad_inputs_fail_loudly():
    from textreact_amd.tanimoto import TanimotoIndex, TrxTanimotoError
    rng = np.random.default_rng(6)
    corpus = fingerprints(rng, 64 * 5 + 9, 512)
    queries = fingerprints(rng, 4, 512)
    idx = TanimotoIndex(512)
    idx.add(corpus[:128]); idx.add(corpus[128:320]); idx.add(corpus[320:])
    sim, rank = idx.search(queries, 50)
    want_s, want_r = oracle.search(queries, corpus, 50)
    assert np.array_equal(rank.cpu().numpy(), want_r) and np.array_equal(sim.cpu().numpy(), want_s)
    idx.add(corpus[:3])                                         # the last block is partially filled: it is packed again
    idx.add(corpus[3:70].astype(np.int32)); idx.add(corpus[70:71])
    grown = np.concatenate([corpus, corpus[:71]])
    sim, rank = idx.search(queries, 50)
    want_s, want_r = oracle.search(queries, grown, 50)
    assert idx.ntotal == len(grown) and np.array_equal(rank.cpu().numpy(), want_r) and np.array_equal(sim.cpu().numpy(), want_s)
    big = corpus[:64].copy(); big[3, 5] = 300
    with pytest.raises(TrxTanimotoError):
        idx.add(big)                                            # rejected, and the index is left as it was
    sim, rank = idx.search(queries, 50)
    assert idx.ntotal == len(grown) and np.array_equal(rank.cpu().numpy(), want_r) and np.array_equal(sim.cpu().numpy(), want_s)
    with pytest.raises(TrxTanimotoError):
        TanimotoIndex(512).add(big)
    with pytest.raises(TrxTanimotoError):
        TanimotoIndex(512).add(np.full((64, 512), 200, dtype=np.int64))       # sum of magnitudes >= 32768
    with pytest.raises(TrxTanimotoError):
        TanimotoIndex(512).add(np.zeros((64, 512), dtype=np.float32))
    with pytest.raises(TrxTanimotoError):
        idx.search(big[:4], 5)


def test_command_line_writes_the_reference_json(tmp_path):
    import json
    from textreact_amd import tanimoto
    rng = np.random.default_rng(9)
    corpus, queries = fingerprints(rng, 300, 2048), fingerprints(rng, 7, 2048)
    np.save(tmp_path / "train.npy", corpus); np.save(tmp_path / "test.npy", queries)
    tanimoto.main(["--train_fps", str(tmp_path / "train.npy"), "--test_fps", str(tmp_path / "test.npy"), "--output",
                   str(tmp_path / "test_nn.json"), "--limit", "5"])
    got = json.load(open(tmp_path / "test_nn.json"))
    want_s, want_r = oracle.search(queries[:5], corpus, 100)
    assert sorted(got) == ["0", "1", "2", "3", "4"]                     # json.dump turns the row numbers into strings
    assert got["3"]["rank"] == want_r[3].tolist() and got["3"]["similarity"] == want_s[3].tolist()


def test_sharded_command_line_writes_the_file_one_gpu_writes(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m textreact_amd.tanimoto ...: the train rows split over two ranks (both
    on this box's GPU, gloo transport), the HIP Tanimoto index on each, one all-gather of keys; rank 0's test_nn.json is byte
    for byte the single-GPU file, ties across the shard boundary included"""
    import subprocess
    import sys
    from textreact_amd import tanimoto
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(11)
    base = fingerprints(rng, 40, 2048)
    corpus = base[rng.integers(0, 40, 901)]               # many equal similarities, spread over both shards
    queries = base[:9]
    np.save(tmp_path / "train.npy", corpus); np.save(tmp_path / "test.npy", queries)
    argv = ["--train_fps", str(tmp_path / "train.npy"), "--test_fps", str(tmp_path / "test.npy"), "--limit", "-1"]
    tanimoto.main(argv + ["--output", str(tmp_path / "one.json")])
    env = dict(os.environ, TRX_DIST_BACKEND="gloo", TRX_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29741", "-m", "textreact_amd.tanimoto"] + argv + ["--output", str(tmp_path / "two.json")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert (tmp_path / "two.json").read_bytes() == (tmp_path / "one.json").read_bytes()


def test_single_rank_sharded_wrapper_applies_the_row_offset():
    from textreact_amd.tanimoto import ShardedTanimotoIndex
    rng = np.random.default_rng(10)
    corpus, queries = fingerprints(rng, 700, 1024), fingerprints(rng, 6, 1024)
    idx = ShardedTanimotoIndex(1024)
    idx.add_shard(corpus, 4096, 4096 + 700)
    sim, rank = idx.search(queries, 50)
    want_s, want_r = oracle.search(queries, corpus, 50)
    assert np.array_equal(rank.cpu().numpy(), want_r + 4096) and np.array_equal(sim.cpu().numpy(), want_s)


def _tani_rank_worker(rank, world, port, ret):
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_tanimoto_cpu import fingerprints as fp_
    from textreact_amd.sharded import shard_bounds
    from textreact_amd.tanimoto import ShardedTanimotoIndex
    rng = np.random.default_rng(11)
    base = fp_(rng, 50, 2048)
    corpus = np.concatenate([fp_(rng, 9000, 2048), base[rng.integers(0, 50, 3001)]])     # ties across the shard boundaries
    corpus = corpus[rng.permutation(len(corpus))]
    queries = np.concatenate([base[:4], fp_(rng, 5, 2048)])
    lo, hi = shard_bounds(len(corpus), world, rank)
    idx = ShardedTanimotoIndex(2048, device=0)
    idx.add_shard(corpus[lo:hi], lo, len(corpus))
    sim, rk = idx.search(queries, 100)
    ret[rank] = (sim.cpu().numpy(), rk.cpu().numpy(), corpus, queries)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_sharded_tanimoto_on_one_gpu(world):
    """the N > 1 path with the real HIP index on every rank (gloo carries the gather: one GPU here)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_tani_rank_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        sim, rk, corpus, queries = ret[r]
        want_s, want_r = oracle.search(queries, corpus, 100)
        assert np.array_equal(rk, want_r), r
        assert np.array_equal(sim, want_s), r
