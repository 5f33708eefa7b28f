"""Generates the committed golden fixtures of tests/golden/.

Run HERE (the build container), never on the GPU box:
    python tests/golden/make_golden.py

  knn_*.npz        small seeded inputs + the oracle's answers (both modes).  The reference holds no
                   fixtures for this path and FAISS cannot run here, so these
                   pin the ORACLE against regressions and give the GPU tests fixed vectors; they do
                   not pin FAISS (the oracle header says "parity unpinned").
  neighbors_*.json outputs of the reference's own Python, imported from /root/reference with an
                   rdkit stub: textreact.dataset.BaseDataset.get_neighbor_text / load_corpus
                   (dataset.py:40-80) on a tiny corpus -- these DO pin the neighbor-file contract.
"""
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _data import bf16_round, gaussian, grid, morgan_like, reaction_fp_like  # noqa: E402
from oracle import flat_knn as oracle  # noqa: E402


def knn_case(name, metric, x, y, k):
    Df, If = oracle.knn_faiss(metric, x, y, k)
    Dc, Ic = oracle.knn_canonical(metric, x, y, k)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), metric=metric, k=k, x=x, y=y,
                        D_faiss=Df, I_faiss=If, D_canonical=Dc, I_canonical=Ic)
    print(name, x.shape, y.shape, "faiss==canonical ids:", bool(np.array_equal(If, Ic)))


def make_knn():
    knn_case("knn_grid_ip", 0, grid(24, 64, 2), grid(700, 64, 1), 10)
    knn_case("knn_grid_l2", 1, grid(24, 64, 2), grid(700, 64, 1), 10)
    y = reaction_fp_like(600, 256, 7)
    knn_case("knn_rxnfp_l2_k20", 1, y[:32].copy(), y, 20)
    y = morgan_like(500, 128, 8)
    y[100:110] = y[100]
    knn_case("knn_morgan_l2_k20", 1, y[:32].copy(), y, 20)
    knn_case("knn_gauss_ip", 0, gaussian(24, 96, 5678), gaussian(800, 96, 1234), 10)
    knn_case("knn_gauss_l2", 1, gaussian(24, 96, 5678), gaussian(800, 96, 1234), 10)
    knn_case("knn_gauss_bf16_ip", 0, bf16_round(gaussian(24, 96, 5678)), bf16_round(gaussian(800, 96, 1234)), 10)
    knn_case("knn_small_n", 0, gaussian(21, 16, 1), gaussian(4, 16, 2), 6)
    knn_case("knn_seq_path", 1, gaussian(5, 32, 1), gaussian(300, 32, 2), 7)


def make_neighbors():
    """Import the reference's dataset module (Python) and record what it does with a neighbor file."""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference not mounted: neighbor goldens not regenerated")
        return
    for mod in ("rdkit", "rdkit.Chem"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["rdkit"].Chem = sys.modules["rdkit.Chem"]
    sys.path.insert(0, ref)
    from textreact.dataset import BaseDataset  # noqa: E402

    corpus = {"c%d" % i: "text %d" % (i % 7) for i in range(20)}   # duplicated texts -> dedup path
    nn = [{"id": "q%d" % i, "nn": ["c%d" % ((i * 3 + j * 5) % 24) for j in range(8)]} for i in range(6)]
    nn[2]["nn"][0] = "q2"      # gold neighbour present in the list
    corpus["q2"] = "gold text"
    nn_path = os.path.join(HERE, "neighbors_input.json")
    with open(nn_path, "w") as f:
        json.dump(nn, f)

    class Args:
        use_gold_neighbor = False
        max_num_neighbors = 10
        random_neighbor_ratio = 0.0
        num_neighbors = 3

    def dataset(split, **kw):
        ds = BaseDataset.__new__(BaseDataset)
        ds.args = Args()
        for k_, v_ in kw.items():
            setattr(ds.args, k_, v_)
        ds.split = split
        ds.indices = [e["id"] for e in nn]
        ds.skip_gold_neighbor = False
        ds.load_corpus(corpus, nn_path)
        return ds

    out = {"corpus": corpus, "cases": []}
    for split, kw, skip in (("train", {}, False), ("train", {"use_gold_neighbor": True}, False),
                            ("test", {}, False), ("test", {}, True),
                            ("train", {"random_neighbor_ratio": 1.0}, False)):
        ds = dataset(split, **kw)
        ds.skip_gold_neighbor = skip
        random.seed(1234)
        lists = [ds.get_neighbor_text(i, return_list=True) for i in range(len(nn))]
        random.seed(1234)
        texts = [ds.get_neighbor_text(i) for i in range(len(nn))]
        out["cases"].append({"split": split, "args": kw, "skip_gold": skip, "lists": lists, "texts": texts,
                             "neighbors": ds.neighbors})
    with open(os.path.join(HERE, "neighbors_expected.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("neighbors goldens written:", len(out["cases"]), "cases")


if __name__ == "__main__":
    make_knn()
    make_neighbors()
