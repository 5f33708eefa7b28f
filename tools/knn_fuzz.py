"""randomised shapes / metrics / input classes through the flat index against the CPU oracle, bit for bit:
python3 tools/knn_fuzz.py [n_cases] [seed]"""
import sys, os, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import textreact_amd.faiss_compat as faiss
from oracle import flat_knn as oracle
from _data import bf16_round, gaussian, grid, morgan_like, reaction_fp_like
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
nfail = 0
for it in range(n_cases):
    metric = rng.choice([0, 1])
    kind = rng.choice(["gauss", "gauss_bf16", "grid", "fp", "dup", "crowd", "crowd"])
    d = rng.choice([1, 3, 17, 64, 65, 100, 128, 200, 256, 300, 768, 1024])
    n = rng.choice([1, 2, 7, 255, 256, 257, 1000, 5000, 20000, 60000])
    nq = rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 700])
    k = rng.choice([1, 2, 10, 20, 24, 25, 64, 100, 256, 300, 2048])      # (25 .. 256: the two-scan path; above: the exact scan)
    if kind == "gauss":
        y, x = gaussian(n, d, 2 * it), gaussian(nq, d, 2 * it + 1)
    elif kind == "gauss_bf16":
        y, x = bf16_round(gaussian(n, d, 2 * it)), bf16_round(gaussian(nq, d, 2 * it + 1))
    elif kind == "grid":
        y, x = grid(n, d, 2 * it), grid(nq, d, 2 * it + 1)
    elif kind == "fp":
        y, x = morgan_like(n, d, 2 * it).astype(np.float32), morgan_like(nq, d, 2 * it + 1).astype(np.float32)
    elif kind == "crowd":   # near-duplicates around the k-th place: crowds of random size that fall off from their queries in steps far
        # below the certificate's slack -- small ones are settled by the wide re-score, wider ones by the fixed-threshold
        # re-scan, the widest by the exact scan (round 4: the three fall-back tiers), mixed with ordinary queries
        y, x = gaussian(n, d, 2 * it), gaussian(nq, d, 2 * it + 1)
        r2 = np.random.default_rng(1000 + it)
        pos = 0
        for c in range(min(nq, r2.integers(1, 6))):
            size = int(min(n - pos, r2.choice([40, 300, 1500, 6000])))
            if size <= 0:
                break
            base = (3.0 * r2.standard_normal(d)).astype(np.float32)
            step = float(r2.choice([1e-7, 1e-8, 3e-6]))
            y[pos:pos + size] = base[None] * (1.0 - step * np.arange(size, dtype=np.float32))[:, None]
            x[c] = base
            pos += size
    else:   # clusters of exact duplicates: ties everywhere
        base = gaussian(max(1, n // 8), d, 2 * it)
        y = base[np.random.default_rng(it).integers(0, base.shape[0], n)]
        x = gaussian(nq, d, 2 * it + 1)
    # round 6: hostile values in a third of the cases -- NaN / +-inf components, rows of float32's largest magnitudes, an all-NaN
    # query -- in random rows of the corpus and of the queries; and the host dtype the arrays are handed over in
    hostile = rng.random() < 0.33
    if hostile:
        r3 = np.random.default_rng(5000 + it)
        for arr, frac in ((y, 0.004), (x, 0.03)):
            for row in r3.choice(arr.shape[0], size=max(1, int(arr.shape[0] * frac)), replace=False):
                what = r3.integers(0, 5)
                col = r3.integers(0, d)
                if what == 0: arr[row, col] = np.nan
                elif what == 1: arr[row, col] = np.inf
                elif what == 2: arr[row, col] = -np.inf
                elif what == 3: arr[row, :] = np.float32(3.0e38) * (1 if r3.random() < 0.5 else -1)
                else: arr[row, :] = np.nan
    as_dtype = None
    if kind in ("fp", "grid") and not hostile and rng.random() < 0.5:      # integer-valued data through the integer host types
        if kind == "grid":
            y, x = np.round(y * 8), np.round(x * 8)
        as_dtype = rng.choice([np.int64, np.int32, np.int16, np.int8])
        y, x = y.astype(np.float32), x.astype(np.float32)
    idx = faiss.IndexFlat(d, metric)
    chunks = rng.choice([1, 1, 3])
    if os.environ.get("TRX_FUZZ_VERBOSE"):
        print("case", it, dict(metric=metric, kind=kind, d=d, n=n, nq=nq, k=k, chunks=chunks), flush=True)
    for part in np.array_split(y, chunks):
        if part.shape[0]:
            idx.add(part if as_dtype is None else part.astype(as_dtype))
    D, I = idx.search(x if as_dtype is None else x.astype(as_dtype), k)
    with np.errstate(all="ignore"):
        Dr, Ir = oracle.knn_canonical(metric, x, y, k)
    ok = np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32))
    if not ok:
        nfail += 1
        bad = np.argwhere(I != Ir)
        print("FAIL", it, dict(metric=metric, kind=kind, d=d, n=n, nq=nq, k=k, chunks=chunks, hostile=hostile, as_dtype=str(as_dtype)), "first bad", bad[:3].tolist(), idx.last_stats())
    tiers = idx.last_stats()
    seen = globals().setdefault("seen", [0, 0, 0])
    seen[0] += tiers["n_rescored"]; seen[1] += tiers["n_rescanned"]; seen[2] += tiers["n_uncertified"] if k <= 24 else 0
print(n_cases, "cases,", nfail, "failures; queries through the wide re-score / the re-scan / the exact scan (k <= 24):", seen)
