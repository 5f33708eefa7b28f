"""Neighbor files: the wire format between retrieval and the predictor.

Format (reference writer retrieve/retrieve_faiss.py:116-118, reader textreact/dataset.py:40-44):
UTF-8 JSON written by ``json.dump(result, f)`` with default separators,
``[{"id": <query id>, "nn": [<corpus id>, ... k items, best first]}, ...]`` in query order.
Also the Tevatron ranking jsonl that the external dense retriever emits and
retrieve/convert_format.py:6-16 turns into the same JSON.
"""
import json


def _py(v):
    """ids come out of pandas / numpy as numpy scalars; json needs plain Python values."""
    return v.item() if hasattr(v, "item") else v


def build_result(query_ids, rank, corpus_ids):
    """[{'id': qid, 'nn': [corpus_ids[n] ...]}] exactly as retrieve_faiss.py:116 builds it.
    Padding entries (-1: fewer than k vectors indexed) are dropped rather than wrapped around."""
    corpus_ids = list(corpus_ids)
    return [{"id": _py(qid), "nn": [_py(corpus_ids[int(n)]) for n in nn if int(n) >= 0]}
            for qid, nn in zip(query_ids, rank)]


def write_neighbors(path, result):
    with open(path, "w") as f:
        json.dump(result, f)  # default separators: byte-identical to the reference's files


def write_neighbor_file(path, query_ids, rank, corpus_ids, block=16384):
    """The same bytes as ``write_neighbors(path, build_result(query_ids, rank, corpus_ids))`` -- ``json.dump`` of the reference's
    list (retrieve/retrieve_faiss.py:116-118), default separators -- without building 14 million Python objects first: every corpus
    id is encoded ONCE by the json encoder (ints, strings with escapes, whatever the 'id' column holds), a row's text is the
    join of its neighbours' encodings, and the file is written a block of rows at a time.  After a 0.4 s search of 680,000
    queries the reference's two lines (list comprehension + json.dump) are 30+ s of the run; this is a few."""
    import numpy as np
    enc = json.JSONEncoder().encode
    tok = np.empty(len(corpus_ids), dtype=object)
    for i, c in enumerate(corpus_ids):
        tok[i] = enc(_py(c))
    rank = np.asarray(rank)
    qids = list(query_ids)
    with open(path, "w") as f:
        f.write("[")
        for r0 in range(0, len(qids), block):
            blk = rank[r0:r0 + block]
            valid = blk >= 0
            toks = tok[np.where(valid, blk, 0)] if len(tok) else np.empty(blk.shape, dtype=object)
            allv = valid.all(axis=1) if blk.ndim == 2 else []
            rows = []
            for j, qid in enumerate(qids[r0:r0 + block]):
                nn = toks[j] if allv[j] else toks[j][valid[j]]
                rows.append('{"id": %s, "nn": [%s]}' % (enc(_py(qid)), ", ".join(nn)))
            if rows:
                f.write((", " if r0 else "") + ", ".join(rows))
        f.write("]")


def read_neighbors(path):
    """{id: nn} as BaseDataset.load_corpus builds it (textreact/dataset.py:42-44)."""
    with open(path) as f:
        return {ex["id"]: ex["nn"] for ex in json.load(f)}


def filter_to_corpus(neighbors, rxn_id, corpus):
    """neighbor ids that exist in the corpus, order kept (textreact/dataset.py:60)."""
    return [i for i in neighbors[rxn_id] if i in corpus]


def convert_tevatron(lines):
    """Tevatron ranking jsonl -> neighbor list (retrieve/convert_format.py:6-16):
    ``{"query_id": q, "negative_passages": [{"docid": d}, ...]}`` per line."""
    out = []
    for line in lines:
        line = line.strip()
        if not line:
            continue
        data = json.loads(line)
        out.append({"id": data["query_id"], "nn": [p["docid"] for p in data["negative_passages"]]})
    return out


def convert_tevatron_file(input_path, output_path):
    with open(input_path) as f:
        out = convert_tevatron(f)
    write_neighbors(output_path, out)
    return len(out)


class NeighborStore:
    """The neighbor-consumer half of the seam: what textreact/dataset.py's BaseDataset does with a
    neighbor file (load_corpus :40-44, deduplicate_neighbors :46-56, get_neighbor_text :58-80),
    kept free of tokenizers / pandas so that the retrieval side can be checked end to end.

    Arguments mirror the reference's ``args`` fields: use_gold_neighbor, max_num_neighbors,
    random_neighbor_ratio, num_neighbors; ``split`` is 'train' or anything else (eval)."""

    def __init__(self, indices, split='train', use_gold_neighbor=False, max_num_neighbors=10,
                 random_neighbor_ratio=0.0, num_neighbors=3, rng=None):
        import random as _random
        self.indices = list(indices)
        self.split = split
        self.use_gold_neighbor = use_gold_neighbor
        self.max_num_neighbors = max_num_neighbors
        self.random_neighbor_ratio = random_neighbor_ratio
        self.num_neighbors = num_neighbors
        self.skip_gold_neighbor = False
        self.corpus = None
        self.neighbors = None
        self._random = rng if rng is not None else _random

    def load_corpus(self, corpus, nn_file):
        self.corpus = corpus
        self.neighbors = read_neighbors(nn_file)

    def deduplicate_neighbors(self, neighbors_ids):
        """keep the first id of every distinct corpus TEXT, order preserved (dataset.py:46-56)."""
        output = []
        for i in neighbors_ids:
            if not any(self.corpus[i] == self.corpus[j] for j in output):
                output.append(i)
        return output

    # -- selection pipeline, one stage per method --------------------------------------------
    def _known_ids(self, rxn_id):
        """retrieved ids that exist in the corpus, retrieval order kept (dataset.py:60)"""
        return [i for i in self.neighbors[rxn_id] if i in self.corpus]

    def _with_gold_first(self, rxn_id, ids):
        """training option: the query's own corpus entry leads the list (dataset.py:62-66)"""
        ids = [i for k, i in enumerate(ids) if not (i == rxn_id and k == ids.index(rxn_id))]
        return ([rxn_id] if rxn_id in self.corpus else []) + ids

    def _without_gold_text(self, rxn_id, ids):
        """evaluation option: drop every neighbour whose TEXT equals the gold text (dataset.py:74-76)"""
        if rxn_id not in self.corpus:
            return ids
        gold = self.corpus[rxn_id]
        return [i for i in ids if self.corpus[i] != gold]

    def select_ids(self, idx):
        """ids whose texts get_neighbor_text would emit, before the random-sample branch."""
        rxn_id = self.indices[idx]
        ids = self._known_ids(rxn_id)
        if self.split == 'train':
            if self.use_gold_neighbor:
                ids = self._with_gold_first(rxn_id, ids)
            return self.deduplicate_neighbors(ids)[:self.max_num_neighbors]
        if self.skip_gold_neighbor:
            ids = self._without_gold_text(rxn_id, ids)
        return self.deduplicate_neighbors(ids)[:self.num_neighbors]

    def get_neighbor_text(self, idx, return_list=False):
        texts = [self.corpus[i] for i in self.select_ids(idx)]
        if self.split == 'train':
            # one draw decides between a random subset and the top-k (dataset.py:69-72); the draw
            # happens even when the ratio is 0, which keeps the RNG stream aligned with the reference
            if self._random.random() < self.random_neighbor_ratio:
                texts = self._random.sample(texts, k=min(self.num_neighbors, len(texts)))
            else:
                texts = texts[:self.num_neighbors]
        if return_list:
            return texts
        return ''.join(' (%d) %s' % (i, t) for i, t in enumerate(texts))
