"""CPU oracle for the Tanimoto brute force (TEST INFRASTRUCTURE ONLY -- the product never imports this module).

Restates, for dense count vectors, what the reference's retrieve/retrieve.py computes:
  :18-31  reaction_similarity -> rdkit.DataStructs.TanimotoSimilarity(fp1, fp2) on two difference fingerprints
  :34-40  one query against every train reaction
  :55-62  ranks = np.argsort(similarities)[::-1][:100]; similarity = [similarities[j] for j in ranks]

PARITY UNPINNED: RDKit is a third-party dependency that is absent from this image (and from /root/reference; the
reference pins no version), so there is no golden vector of the reference itself for this path.  The arithmetic below
is RDKit's published algorithm for sparse count vectors (Code/DataStructs/SparseIntVect.h: TanimotoSimilarity ->
TverskySimilarity(v1, v2, 1, 1) -> calcVectParams): with |v| = sum_i |v_i| (absolute values: difference fingerprints
hold signed counts) and and = sum over the common positions of min(|v1_i|, |v2_i|),
        sim = and / (|v1| + |v2| - and),      0.0 when the denominator is < 1e-6,
in double precision.  `sparse_similarity` walks two sorted sparse vectors the way calcVectParams does;
`similarities` is the vectorised statement the tests use at size; tests check one against the other.

np.argsort's default sort is not stable, so the reference's order among equal similarities is unspecified; the
rule fixed here (and in the HIP path) is the one a stable ascending argsort read backwards gives: among equal
similarities the LARGER row number comes first.
"""
import numpy as np


def sparse_similarity(v1, v2):
    """v1, v2: dicts {position: signed count} (nonzero entries).  calcVectParams + TverskySimilarity(1, 1)."""
    it1, it2 = sorted(v1.items()), sorted(v2.items())
    s1 = float(sum(abs(c) for _, c in it1))
    s2 = float(sum(abs(c) for _, c in it2))
    both, i2 = 0.0, 0
    for pos, c in it1:
        while i2 < len(it2) and it2[i2][0] < pos:
            i2 += 1
        if i2 < len(it2) and it2[i2][0] == pos:
            both += float(min(abs(c), abs(it2[i2][1])))
    den = s1 + s2 - both
    return 0.0 if abs(den) < 1e-6 else both / den


def similarities(query, corpus):
    """query [d], corpus [N, d] integer counts -> float64 [N]"""
    q = np.abs(np.asarray(query, dtype=np.int64))
    c = np.abs(np.asarray(corpus, dtype=np.int64))
    both = np.minimum(c, q[None, :]).sum(axis=1).astype(np.float64)
    den = c.sum(axis=1).astype(np.float64) + float(q.sum()) - both
    out = np.zeros(len(c), dtype=np.float64)
    ok = np.abs(den) >= 1e-6
    out[ok] = both[ok] / den[ok]
    return out


def rank(sims, k):
    """retrieve.py:59 with the tie rule fixed: stable ascending argsort, read backwards, first k"""
    return np.argsort(sims, kind="stable")[::-1][:k]


def search(queries, corpus, k=100):
    """-> (similarity float64 [Q, k'], rank int64 [Q, k']), k' = min(k, N)"""
    queries = np.asarray(queries)
    S, R = [], []
    for q in queries:
        s = similarities(q, corpus)
        r = rank(s, k)
        S.append(s[r]); R.append(r.astype(np.int64))
    return np.array(S, dtype=np.float64).reshape(len(queries), -1), np.array(R, dtype=np.int64).reshape(len(queries), -1)
