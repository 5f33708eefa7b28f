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
  * ``train_fp.pkl`` gets a sidecar (``train_fp.pkl.meta.json``: field, ``--before``, row count, SHA-256 of the id column
    the rows belong to), so a cache is only reused for the rows it was computed from; a cache without one (written by the
    reference) is accepted when its length identifies it (all rows / the filtered rows);
  * launched by ``python -m torch.distributed.run --nproc-per-node G -m textreact_amd.retrieve_faiss ...`` the train
    vectors are row-sharded over the G GPUs (rank r indexes rows ``shard_bounds(n, G, r)``), every search goes through
    ``sharded.ShardedFlatIndex`` (local scan, all-to-all, merge, all-gather over RCCL) and rank 0 writes the same three
    files, byte for byte what one GPU writes (tests/test_knn_gpu.py);
  * RDKit is imported only when fingerprints must be computed, and ``--train_vectors /
    --valid_vectors / --test_vectors`` (``.npy`` of shape [n, d]) feed precomputed vectors
    instead -- that is how the 768-d dense embeddings of the external retriever enter; with
    ``--metric ip --k 10`` this is BASELINE.json's configs[0..1] through the reference's own CLI.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

from . import faiss_compat as faiss
from .neighbors import build_result, write_neighbor_file, write_neighbors

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
    parser.add_argument('--tie_rule', type=str, default=None, choices=['id', 'faiss'],
                        help="order of exactly equal scores: 'id' (smaller row id first, the default) or 'faiss' (the order "
                             "FAISS's heap leaves for --metric ip; include/trx_knn.h TRX_TIES_FAISS)")
    parser.add_argument('--stage_times', action='store_true',
                        help='print one JSON line with the wall-clock seconds of every stage (read csv, load vectors, index.add, each '
                             'search, id mapping, json.dump) to stderr when the run ends')
    parser.add_argument('--replicas', action='store_true',
                        help='under torch.distributed.run: every GPU holds all train vectors and searches 1/G of the queries '
                             '(FAISS IndexReplicas; sharded.ReplicatedFlatIndex) instead of a row shard of the train vectors')
    return parser


def _dist_setup():
    """(rank, world): one process per GPU when launched by torch.distributed.run, else (0, 1).  The group is set up by
    textreact_amd/_dist.py -- set_device, then init_process_group("nccl", device_id=...) -- like every other N > 1 entry point."""
    from . import _dist
    if _dist.world_size() <= 1:
        return 0, 1
    rank, world, _ = _dist.setup()
    return rank, world


class ShardedSearcher:
    """`index.search(x, k)` of retrieve_faiss.py:71 over a corpus row-sharded across the ranks: numpy in, numpy out, the
    same (D, I) on every rank and the same as one flat index over all rows (sharded.ShardedFlatIndex).  Queries go to
    the device in blocks of 65,536 rows: the train split searches itself, and 680 k x 2048 fp32 queries are 5.6 GB."""

    QUERY_BLOCK = 65536

    def __init__(self, local_rows, lo, ntotal, metric, rank, world):
        from .sharded import ShardedFlatIndex
        d = local_rows.shape[1]
        local = faiss.IndexFlatL2(d) if metric == 'l2' else faiss.IndexFlatIP(d)
        self.device = getattr(local, "device", None)          # the HIP index names its GPU; a CPU stand-in (tests) has none
        self.index = ShardedFlatIndex(d, 1 if metric == 'l2' else 0, local_index=local, merge=getattr(faiss, "merge_topk", None))
        self.index.add_shard(self._tensor(local_rows), lo, ntotal)

    def _tensor(self, x):
        import torch
        x = np.asarray(x)
        if self.device is not None and x.dtype in (np.int8, np.bool_):      # Morgan bit vectors: bytes over PCIe, bf16 (exact) on the GPU
            return torch.from_numpy(np.ascontiguousarray(x).view(np.int8)).cuda(self.device).to(torch.bfloat16)
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return t.cuda(self.device) if self.device is not None else t

    def search(self, x, k):
        D, I = [], []
        for q0 in range(0, len(x), self.QUERY_BLOCK):
            d_, i_ = self.index.search(self._tensor(x[q0:q0 + self.QUERY_BLOCK]), k)
            D.append(d_.cpu().numpy()); I.append(i_.cpu().numpy())
        if not D:
            return np.empty((0, k), np.float32), np.empty((0, k), np.int64)
        return np.concatenate(D), np.concatenate(I)


class ReplicaSearcher(ShardedSearcher):
    """the same protocol over sharded.ReplicatedFlatIndex: all rows on every rank, a G-th of each query block per rank (--replicas)"""

    def __init__(self, rows, metric, rank, world):
        from .sharded import ReplicatedFlatIndex
        d = rows.shape[1]
        local = faiss.IndexFlatL2(d) if metric == 'l2' else faiss.IndexFlatIP(d)
        self.device = getattr(local, "device", None)
        self.index = ReplicatedFlatIndex(d, 1 if metric == 'l2' else 0, local_index=local)
        for r0 in range(0, len(rows), self.QUERY_BLOCK):      # (in blocks: the fp32 staging copy of 680 k x 2048 rows is 5.6 GB)
            self.index.add(self._tensor(rows[r0:r0 + self.QUERY_BLOCK]))


def _ids_digest(ids):
    import hashlib
    h = hashlib.sha256()
    for v in ids:
        h.update(str(v).encode("utf-8")); h.update(b"\0")
    return h.hexdigest()


def load_or_compute_train_fps(train_df, keep, args, fingerprint_fn, train_fp_file):
    """the `train_fp.pkl` cache (retrieve_faiss.py:100-110; NumPy .npy bytes despite the name) -> the fingerprints of the
    rows `keep` selects (None = all).  The reference reuses whatever file is there (:101-103,:112); here a sidecar says
    which rows a cache holds, and a cache that does not belong to these rows is recomputed, not trusted."""
    meta_file = train_fp_file + ".meta.json"
    kept_df = train_df[keep].reset_index(drop=True) if keep is not None else train_df
    want = {"field": args.field, "before": int(args.before), "rows": int(len(kept_df)), "ids_sha256": _ids_digest(kept_df['id'])}
    if os.path.exists(train_fp_file):
        with open(train_fp_file, 'rb') as f:
            fps = np.load(f)
        meta = None
        if os.path.exists(meta_file):
            with open(meta_file) as f:
                meta = json.load(f)
        if meta is not None:
            if all(meta.get(k_) == want[k_] for k_ in ("field", "rows", "ids_sha256")) and len(fps) == want["rows"]:
                return fps                                   # computed from exactly these rows
            if (keep is not None and meta.get("field") == args.field and meta.get("before") == -1 and len(fps) == len(train_df)
                    and meta.get("ids_sha256") == _ids_digest(train_df['id'])):
                return fps[keep]                             # computed from all rows of this file: filter it like the ids
            print("train_fp.pkl was computed from other rows (%s): recomputing" % json.dumps(meta))
        elif keep is not None and len(fps) == len(train_df):
            return fps[keep]       # no sidecar (a reference-written cache), one row per row of the file: unfiltered
        elif len(fps) == len(kept_df):
            return fps             # no sidecar, one row per kept row (the reference wrote it under the same --before)
        else:
            print("train_fp.pkl holds %d rows, the train file %d (%d after --before): recomputing" % (len(fps), len(train_df), len(kept_df)))
    fps = fingerprint_fn(kept_df[args.field])
    with open(train_fp_file, 'wb') as f:
        np.save(f, fps)
    with open(meta_file, 'w') as f:
        json.dump(want, f)
    return fps


def main(argv=None):
    import pandas as pd
    args = get_parser().parse_args(argv)
    if args.tie_rule is not None:
        os.environ["TRX_TIE_RULE"] = args.tie_rule         # read by every index faiss_compat makes from here on
    rank, world = _dist_setup()
    say = print if rank == 0 else (lambda *a, **k: None)
    stages, t_last = {}, [time.perf_counter()]

    def lap(name):      # seconds since the previous lap, added to stage `name`
        now = time.perf_counter()
        stages[name] = stages.get(name, 0.0) + now - t_last[0]
        t_last[0] = now

    train_df = pd.read_csv(os.path.join(args.data_path, args.train_file), keep_default_na=False)
    val_df = pd.read_csv(os.path.join(args.data_path, args.valid_file), keep_default_na=False)
    test_df = pd.read_csv(os.path.join(args.data_path, args.test_file), keep_default_na=False)

    lap("read_csv")
    if args.field == 'canonical_rxn':
        say('Reaction fingerprint')
        fingerprint_fn = compute_reaction_fingerprints
    else:
        say('Molecule fingerprint')
        fingerprint_fn = compute_molecule_fingerprints

    if args.before != -1:  # unconditional: see module docstring
        keep = (train_df['year'] < args.before).to_numpy()
    else:
        keep = None

    if rank == 0:
        os.makedirs(args.output_path, exist_ok=True)
    train_fp_file = os.path.join(args.output_path, 'train_fp.pkl')
    if args.train_vectors:
        train_fps = np.load(args.train_vectors, mmap_mode='r' if world > 1 else None)   # sharded: a rank touches its rows only
        if keep is not None:
            train_fps = train_fps[keep] if world == 1 else _RowView(train_fps, np.flatnonzero(keep))
    else:
        if rank == 0:      # one rank computes (a 64-process RDKit pool) and writes the cache; the others read it
            train_fps = load_or_compute_train_fps(train_df, keep, args, fingerprint_fn, train_fp_file)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            if rank != 0:
                train_fps = load_or_compute_train_fps(train_df, keep, args, fingerprint_fn, train_fp_file)
    if keep is not None:
        train_df = train_df[keep].reset_index(drop=True)
    assert len(train_fps) == len(train_df), "fingerprints and train ids are misaligned"
    train_id = train_df['id']

    lap("load_train_vectors")
    say('Faiss build index')
    if world == 1:
        index = build_index(train_fps, args.metric)
    else:
        from .sharded import shard_bounds
        lo, hi = shard_bounds(len(train_fps), world, rank)
        index = (ReplicaSearcher(train_fps, args.metric, rank, world) if args.replicas else
                 ShardedSearcher(np.asarray(train_fps[lo:hi]), lo, len(train_fps), args.metric, rank, world))

    def vectors(df, path):
        if path:
            return np.load(path)
        if world == 1:
            return fingerprint_fn(df[args.field])
        import torch.distributed as dist
        box = [fingerprint_fn(df[args.field]) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    lap("index_add")
    rank_arr = None
    for name, df, fps in (('train', train_df, train_fps),
                          ('val', val_df, None),
                          ('test', test_df, None)):
        if fps is None:
            fps = vectors(df, args.valid_vectors if name == 'val' else args.test_vectors)
        lap("load_query_vectors")
        if rank == 0:
            rank_arr = index_and_search(train_fps, fps, k=args.k, metric=args.metric, index=index)
            lap("search_" + name)
            write_neighbor_file(os.path.join(args.output_path, name + '.json'), df['id'], rank_arr, train_id)
            lap("map_ids_and_write_json")
        else:
            index.search(fps, args.k)
            lap("search_" + name)

    if rank == 0 and args.field == 'canonical_rxn' and all(f in test_df.columns for f in CONDITION_FIELDS):
        cnt = hit_rates(rank_arr, test_df, train_df)
        print(cnt, len(test_df))
        for x in cnt:
            print(f"Top-{x}: {cnt[x] / len(test_df):.4f}", end='  ')
        print()
    lap("hit_rates")
    if args.stage_times and rank == 0:
        stages["total"] = sum(stages.values())
        print(json.dumps({"stage_seconds": {k_: round(v, 3) for k_, v in stages.items()}, "train_rows": int(len(train_df)), "k": args.k,
                          "queries": {"train": int(len(train_df)), "val": int(len(val_df)), "test": int(len(test_df))}}), file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


class _RowView:
    """rows `rows` of a memory-mapped matrix, read when they are sliced (a rank of the sharded run touches its shard and
    the query blocks, never the whole file at once)"""

    def __init__(self, base, rows):
        self.base, self.rows, self.shape = base, rows, (len(rows), base.shape[1])

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, sl):
        return np.asarray(self.base[self.rows[sl]])


if __name__ == '__main__':
    raise SystemExit(main())
