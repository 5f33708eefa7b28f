"""Drop-in for the reference's retrieval CLI, retrieve/retrieve_faiss.py, on the gfx950 flat index.

Same flags (retrieve_faiss.py:79-87), same outputs (``train.json``, ``val.json``, ``test.json`` in
``--output_path`` plus the ``train_fp.pkl`` fingerprint cache, which -- as in the reference,
:100-110 -- holds NumPy ``.npy`` bytes despite its name), same prints, same hit-rate report for
``--field canonical_rxn`` (:132-144).  ``index_and_search`` keeps the reference signature
(:62-74): IndexFlatL2, k = 20, returns the rank array.

Differences, all deliberate:
  * the corpus is added to the index ONCE and searched three times (the reference rebuilds the
    index and re-adds the corpus for every split, :115/:121/:127);
  * ``--before`` filters ``train_df`` unconditionally; the reference skips the filter when the
    fingerprint cache exists, which misaligns ids and fingerprints (:101-103,:112);
  * RDKit is imported only when fingerprints must be computed, and ``--train_vectors /
    --valid_vectors / --test_vectors`` (``.npy`` of shape [n, d]) feed precomputed vectors
    instead -- that is how the 768-d dense embeddings of the external retriever enter; with
    ``--metric ip --k 10`` this is BASELINE.json's configs[0..1] through the reference's own CLI.
"""
import argparse
import json
import os
import time

import numpy as np

from . import faiss_compat as faiss
from .neighbors import build_result, write_neighbors

CONDITION_FIELDS = ['catalyst1', 'solvent1', 'solvent2', 'reagent1', 'reagent2']


# ---- fingerprints (RDKit; retrieve_faiss.py:18-50) ---------------------------------------------
def _rdkit():
    import rdkit  # noqa: F401
    import rdkit.Chem as Chem
    import rdkit.Chem.AllChem as AllChem
    import rdkit.Chem.rdChemReactions as rdChemReactions
    import rdkit.DataStructs as DataStructs
    rdkit.RDLogger.DisableLog('rdApp.*')
    return Chem, AllChem, rdChemReactions, DataStructs


def reaction_fingerprint_array(smiles):
    _, _, rdChemReactions, _ = _rdkit()
    rxn = rdChemReactions.ReactionFromSmarts(smiles)
    fp = rdChemReactions.CreateDifferenceFingerprintForReaction(rxn)
    return np.array([x for x in fp])


def morgan_fingerprint(smiles):
    Chem, AllChem, _, DataStructs = _rdkit()
    try:
        mol = Chem.MolFromSmiles(smiles)
        fp = AllChem.GetMorganFingerprintAsBitVect(mol, 2, nBits=1024)
        array = np.zeros((0,), dtype=np.int8)
        DataStructs.ConvertToNumpyArray(fp, array)
    except Exception:
        return morgan_fingerprint('C')
    return array


def _pool_map(fn, items, chunksize, num_workers):
    import multiprocessing
    with multiprocessing.Pool(num_workers) as p:
        return p.map(fn, items, chunksize=chunksize)


def compute_reaction_fingerprints(smiles_list, num_workers=64):
    return np.array(_pool_map(reaction_fingerprint_array, smiles_list, 128, num_workers))


def compute_molecule_fingerprints(smiles_list, num_workers=64):
    return np.array(_pool_map(morgan_fingerprint, smiles_list, 64, num_workers))


def compare_condition(row1, row2):
    """True when the five condition fields agree, NaN == NaN (retrieve_faiss.py:53-59)."""
    for field in CONDITION_FIELDS:
        if type(row1[field]) is not str and type(row2[field]) is not str:
            continue
        if row1[field] != row2[field]:
            return False
    return True


# ---- the hot path ------------------------------------------------------------------------------
def build_index(train_fps, metric='l2'):
    d = train_fps.shape[1]
    index = faiss.IndexFlatL2(d) if metric == 'l2' else faiss.IndexFlatIP(d)
    index.add(train_fps)
    return index


def index_and_search(train_fps, query_fps, k=20, metric='l2', index=None):
    """retrieve_faiss.py:62-74.  Pass `index` (from build_index) to reuse the corpus already in HBM."""
    if index is None:
        print('Faiss build index')
        index = build_index(train_fps, metric)
    print('Faiss nearest neighbor search')
    start = time.time()
    distance, rank = index.search(query_fps, k)
    end = time.time()
    print(f"{end - start:.2f} s")
    return rank


def hit_rates(rank, test_df, train_df, cutoffs=(1, 3, 5, 10, 15)):
    """condition hit-rate of the retrieved neighbours at the reference's cut-offs (:132-144)."""
    cnt = {x: 0 for x in cutoffs}
    for i, nn in enumerate(rank):
        test_row = test_df.iloc[i]
        hit_map = [compare_condition(test_row, train_df.iloc[n]) for n in nn if n >= 0]
        for x in cnt:
            cnt[x] += bool(np.any(hit_map[:x]))
    return cnt


def get_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--data_path', type=str, default=None, required=True)
    parser.add_argument('--train_file', type=str, default=None, required=True)
    parser.add_argument('--valid_file', type=str, default=None, required=True)
    parser.add_argument('--test_file', type=str, default=None, required=True)
    parser.add_argument('--field', type=str, default='canonical_rxn')
    parser.add_argument('--before', type=int, default=-1)
    parser.add_argument('--output_path', type=str, default=None, required=True)
    # extensions (not in the reference)
    parser.add_argument('--train_vectors', type=str, default=None, help='.npy [n_train, d] instead of fingerprints')
    parser.add_argument('--valid_vectors', type=str, default=None)
    parser.add_argument('--test_vectors', type=str, default=None)
    parser.add_argument('--metric', type=str, default='l2', choices=['l2', 'ip'])
    parser.add_argument('--k', type=int, default=20)
    return parser


def main(argv=None):
    import pandas as pd
    args = get_parser().parse_args(argv)

    train_df = pd.read_csv(os.path.join(args.data_path, args.train_file), keep_default_na=False)
    val_df = pd.read_csv(os.path.join(args.data_path, args.valid_file), keep_default_na=False)
    test_df = pd.read_csv(os.path.join(args.data_path, args.test_file), keep_default_na=False)

    if args.field == 'canonical_rxn':
        print('Reaction fingerprint')
        fingerprint_fn = compute_reaction_fingerprints
    else:
        print('Molecule fingerprint')
        fingerprint_fn = compute_molecule_fingerprints

    if args.before != -1:  # unconditional: see module docstring
        keep = (train_df['year'] < args.before).to_numpy()
    else:
        keep = None

    os.makedirs(args.output_path, exist_ok=True)
    train_fp_file = os.path.join(args.output_path, 'train_fp.pkl')
    if args.train_vectors:
        train_fps = np.load(args.train_vectors)
        if keep is not None:
            train_fps = train_fps[keep]
    elif os.path.exists(train_fp_file):
        with open(train_fp_file, 'rb') as f:
            train_fps = np.load(f)
        if keep is not None and len(train_fps) == len(keep):
            train_fps = train_fps[keep]   # cache written without the filter
    else:
        df = train_df[keep].reset_index(drop=True) if keep is not None else train_df
        train_fps = fingerprint_fn(df[args.field])
        with open(train_fp_file, 'wb') as f:
            np.save(f, train_fps)
    if keep is not None:
        train_df = train_df[keep].reset_index(drop=True)
    assert len(train_fps) == len(train_df), "fingerprints and train ids are misaligned"
    train_id = train_df['id']

    print('Faiss build index')
    index = build_index(train_fps, args.metric)

    def vectors(df, path):
        return np.load(path) if path else fingerprint_fn(df[args.field])

    rank = None
    for name, df, fps in (('train', train_df, train_fps),
                          ('val', val_df, None),
                          ('test', test_df, None)):
        if fps is None:
            fps = vectors(df, args.valid_vectors if name == 'val' else args.test_vectors)
        rank = index_and_search(train_fps, fps, k=args.k, metric=args.metric, index=index)
        result = build_result(df['id'], rank, train_id)
        write_neighbors(os.path.join(args.output_path, name + '.json'), result)

    if args.field == 'canonical_rxn' and all(f in test_df.columns for f in CONDITION_FIELDS):
        cnt = hit_rates(rank, test_df, train_df)
        print(cnt, len(test_df))
        for x in cnt:
            print(f"Top-{x}: {cnt[x] / len(test_df):.4f}", end='  ')
        print()
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
