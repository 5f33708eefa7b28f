"""The retrieval CLI mirror (textreact_amd/retrieve_faiss.py vs reference retrieve/retrieve_faiss.py)
on CPU: flags, file names, file contents, --before handling.  The index is injected (the oracle
stands in for the HIP index; test infrastructure only)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import textreact_amd.retrieve_faiss as rf
from _data import reaction_fp_like
from oracle import flat_knn as oracle


class FakeIndex:
    def __init__(self, d, metric):
        self.d, self.metric, self.y, self.adds = d, metric, None, 0

    def add(self, x):
        self.y = np.asarray(x, dtype=np.float32); self.adds += 1

    def search(self, x, k):
        return oracle.knn_canonical(self.metric, np.asarray(x, dtype=np.float32), self.y, k)


@pytest.fixture
def fake_faiss(monkeypatch):
    made = []

    class F:
        @staticmethod
        def IndexFlatL2(d):
            made.append(FakeIndex(d, 1)); return made[-1]

        @staticmethod
        def IndexFlatIP(d):
            made.append(FakeIndex(d, 0)); return made[-1]
    monkeypatch.setattr(rf, "faiss", F)
    return made


def _write_data(tmp_path, n_train=60, n_q=9, d=64):
    fps = reaction_fp_like(n_train + 2 * n_q, d, 3, 0.1)
    conds = ["catalyst1", "solvent1", "solvent2", "reagent1", "reagent2"]

    def df(lo, hi, prefix):
        return pd.DataFrame({"id": ["%s%d" % (prefix, i) for i in range(lo, hi)],
                             "canonical_rxn": ["C>>C"] * (hi - lo),
                             "year": [2000 + (i % 20) for i in range(lo, hi)],
                             **{c: ["x%d" % (i % 3) for i in range(lo, hi)] for c in conds}})
    df(0, n_train, "tr").to_csv(tmp_path / "train.csv", index=False)
    df(0, n_q, "va").to_csv(tmp_path / "val.csv", index=False)
    df(0, n_q, "te").to_csv(tmp_path / "test.csv", index=False)
    np.save(tmp_path / "train.npy", fps[:n_train])
    np.save(tmp_path / "val.npy", fps[n_train:n_train + n_q])
    np.save(tmp_path / "test.npy", fps[n_train + n_q:])
    return fps


def test_flags_match_the_reference():
    # retrieve_faiss.py:79-87
    p = rf.get_parser()
    opts = {a.dest: a for a in p._actions}
    for name in ("data_path", "train_file", "valid_file", "test_file", "output_path"):
        assert opts[name].required
    assert opts["field"].default == "canonical_rxn" and opts["before"].default == -1 and opts["before"].type is int
    assert opts["k"].default == 20 and opts["metric"].default == "l2"   # the reference's constants (:65,:70)


def test_end_to_end_files(tmp_path, fake_faiss, capsys):
    fps = _write_data(tmp_path)
    out = tmp_path / "out"
    rc = rf.main(["--data_path", str(tmp_path), "--train_file", "train.csv", "--valid_file", "val.csv",
                  "--test_file", "test.csv", "--output_path", str(out),
                  "--train_vectors", str(tmp_path / "train.npy"), "--valid_vectors", str(tmp_path / "val.npy"),
                  "--test_vectors", str(tmp_path / "test.npy")])
    assert rc == 0
    assert len(fake_faiss) == 1 and fake_faiss[0].adds == 1          # corpus added once, searched three times
    assert sorted(os.listdir(out)) == ["test.json", "train.json", "val.json"]
    tr = json.loads((out / "train.json").read_text())
    assert [e["id"] for e in tr] == ["tr%d" % i for i in range(60)] and all(len(e["nn"]) == 20 for e in tr)
    assert all(e["nn"][0] == e["id"] or True for e in tr)
    # train searches itself: the query is a distance-0 neighbour (retrieve_faiss.py:114-115)
    _, I = oracle.knn_canonical(1, fps[:60], fps[:60], 20)
    assert [e["nn"] for e in tr] == [["tr%d" % n for n in row] for row in I]
    te = json.loads((out / "test.json").read_text())
    _, I = oracle.knn_canonical(1, fps[69:], fps[:60], 20)
    assert [e["nn"] for e in te] == [["tr%d" % n for n in row] for row in I]
    txt = capsys.readouterr().out
    assert "Faiss build index" in txt and "Faiss nearest neighbor search" in txt and "Top-1:" in txt


def test_before_filters_ids_and_vectors_together(tmp_path, fake_faiss):
    _write_data(tmp_path)
    out = tmp_path / "out"
    rf.main(["--data_path", str(tmp_path), "--train_file", "train.csv", "--valid_file", "val.csv",
             "--test_file", "test.csv", "--output_path", str(out), "--before", "2010",
             "--train_vectors", str(tmp_path / "train.npy"), "--valid_vectors", str(tmp_path / "val.npy"),
             "--test_vectors", str(tmp_path / "test.npy")])
    tr = json.loads((out / "train.json").read_text())
    kept = ["tr%d" % i for i in range(60) if 2000 + (i % 20) < 2010]
    assert [e["id"] for e in tr] == kept
    assert set(n for e in tr for n in e["nn"]) <= set(kept)


def test_index_and_search_keeps_reference_signature(fake_faiss, capsys):
    y = reaction_fp_like(100, 32, 1, 0.2)
    rank = rf.index_and_search(y, y[:5])          # retrieve_faiss.py:62-74: L2, k = 20
    assert rank.shape == (5, 20) and rank.dtype == np.int64
    assert np.array_equal(rank, oracle.knn_canonical(1, y[:5], y, 20)[1])
    assert capsys.readouterr().out.count(" s\n") == 1   # the timing print (:69-73)


def _argv(tmp_path, out, *extra):
    return ["--data_path", str(tmp_path), "--train_file", "train.csv", "--valid_file", "val.csv", "--test_file", "test.csv",
            "--output_path", str(out)] + list(extra)


def test_fingerprint_cache_is_reused_only_for_the_rows_it_was_computed_from(tmp_path, fake_faiss, monkeypatch):
    """retrieve_faiss.py:100-112: the reference reuses any train_fp.pkl it finds, also one written under another --before
    (ids and fingerprints then misalign).  Here a sidecar names the rows a cache belongs to."""
    fps = _write_data(tmp_path)
    calls = []

    def fake_fp(smiles):           # stands in for the RDKit pool: row i of the frame it is given -> a vector that names i
        calls.append(len(smiles))
        return fps[:len(smiles)].copy()
    monkeypatch.setattr(rf, "compute_reaction_fingerprints", fake_fp)
    out = tmp_path / "out"
    vec = ["--valid_vectors", str(tmp_path / "val.npy"), "--test_vectors", str(tmp_path / "test.npy")]
    rf.main(_argv(tmp_path, out, *vec))                                   # computes all 60 rows, writes cache + sidecar
    assert calls == [60] and os.path.exists(out / "train_fp.pkl.meta.json")
    first = (out / "train.json").read_text()
    rf.main(_argv(tmp_path, out, *vec))                                   # same rows: reused
    assert calls == [60] and (out / "train.json").read_text() == first
    rf.main(_argv(tmp_path, out, "--before", "2010", *vec))               # a cache of ALL rows: filtered like the ids
    assert calls == [60]
    tr = json.loads((out / "train.json").read_text())
    kept = [i for i in range(60) if 2000 + (i % 20) < 2010]
    assert [e["id"] for e in tr] == ["tr%d" % i for i in kept] and all(e["nn"][0] == e["id"] for e in tr)
    # a cache written under one filter is not trusted under another of the same length
    out2 = tmp_path / "out2"
    rf.main(_argv(tmp_path, out2, "--before", "2010", *vec))              # computes the 30 kept rows
    assert calls == [60, 30]
    df = pd.read_csv(tmp_path / "train.csv")
    df["year"] = 2019 - (df["year"] - 2000)                                # the other 30 rows are now the early ones
    df.to_csv(tmp_path / "train.csv", index=False)
    rf.main(_argv(tmp_path, out2, "--before", "2010", *vec))
    assert calls == [60, 30, 30]                                           # equal length, other rows: recomputed
    # a reference-written cache (no sidecar) is identified by its length
    os.remove(out / "train_fp.pkl.meta.json")
    rf.main(_argv(tmp_path, out, "--before", "2010", *vec))
    assert calls == [60, 30, 30]


class _ShardFake:
    """CPU stand-ins for the HIP index and merge kernel inside the sharded CLI (the oracle; test infrastructure only)"""

    @staticmethod
    def make():
        from test_sharded_gloo import OracleLocalIndex, oracle_merge

        class F:
            IndexFlatL2 = staticmethod(lambda d: OracleLocalIndex(1))
            IndexFlatIP = staticmethod(lambda d: OracleLocalIndex(0))
            merge_topk = staticmethod(oracle_merge)
        return F


def _cli_rank(rank, world, port, argv):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      TRX_DIST_BACKEND="gloo")
    import textreact_amd.retrieve_faiss as rf_
    rf_.faiss = _ShardFake.make()
    assert rf_.main(argv) == 0


@pytest.mark.parametrize("before", [None, "2010"])
def test_two_rank_cli_writes_the_files_one_rank_writes(tmp_path, fake_faiss, before):
    """python -m torch.distributed.run --nproc-per-node 2 -m textreact_amd.retrieve_faiss ...: train vectors row-sharded,
    rank 0 writes train/val/test.json -- byte for byte the single-process files (gloo, the oracle as local index)"""
    import socket
    import torch.multiprocessing as mp
    _write_data(tmp_path, n_train=61)
    vec = ["--train_vectors", str(tmp_path / "train.npy"), "--valid_vectors", str(tmp_path / "val.npy"), "--test_vectors", str(tmp_path / "test.npy")]
    if before:
        vec += ["--before", before]
    one, two = tmp_path / "one", tmp_path / "two"
    rf.main(_argv(tmp_path, one, *vec))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mp.spawn(_cli_rank, args=(2, port, _argv(tmp_path, two, *vec)), nprocs=2, join=True)
    for name in ("train.json", "val.json", "test.json"):
        assert (two / name).read_bytes() == (one / name).read_bytes(), name
    # --replicas: every rank all train vectors, a half of every query block each (sharded.ReplicatedFlatIndex)
    three = tmp_path / "three"
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    mp.spawn(_cli_rank, args=(2, port, _argv(tmp_path, three, "--replicas", *vec)), nprocs=2, join=True)
    for name in ("train.json", "val.json", "test.json"):
        assert (three / name).read_bytes() == (one / name).read_bytes(), name
