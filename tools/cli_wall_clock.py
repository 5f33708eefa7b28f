"""End-to-end wall clock of the retrieval CLI at the reference's size (retrieve/retrieve_faiss.py:112-130): 680,000 train rows of
2048-d count fingerprints, 2 x 40,000 val / test queries, k = 20, L2 -- `python -m textreact_amd.retrieve_faiss --stage_times`
with --*_vectors files (RDKit is absent, so the fingerprints are synthetic: the class of retrieve_faiss.py:24-27).
    python3 tools/cli_wall_clock.py [n_train [n_query [dtype]]]   dtype: int8 | int64 (the .npy files' dtype)
Prints one JSON line: the CLI's own stage table, the process wall clock around it, and what the reference's two lines
(list comprehension + json.dump, retrieve_faiss.py:116-118) take on the same rank array, for comparison."""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pandas as pd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 680000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
dt = sys.argv[3] if len(sys.argv) > 3 else "int8"
d = 2048
tmp = tempfile.mkdtemp(prefix="trx_cli_")
try:
    rng = np.random.default_rng(0)
    def fps(m, seed):
        r = np.random.default_rng(seed)
        out = np.zeros((m, d), dtype=dt)
        for r0 in range(0, m, 65536):
            mm = min(65536, m - r0)
            out[r0:r0 + mm] = ((r.random((mm, d), dtype=np.float32) < 0.02) * r.integers(-10, 11, (mm, d))).astype(dt)
        return out
    t0 = time.perf_counter()
    for name, m, seed in (("train", n, 1), ("val", nq, 2), ("test", nq, 3)):
        np.save(os.path.join(tmp, name + ".npy"), fps(m, seed))
        pd.DataFrame({"id": ["US%08d" % i for i in range(m)] if name == "train" else np.arange(m) + (10 ** 7 if name == "val" else 2 * 10 ** 7),
                      "canonical_rxn": ["C>>C"] * m}).to_csv(os.path.join(tmp, name + ".csv"), index=False)
    t_gen = time.perf_counter() - t0
    argv = [sys.executable, "-m", "textreact_amd.retrieve_faiss", "--data_path", tmp, "--train_file", "train.csv", "--valid_file", "val.csv",
            "--test_file", "test.csv", "--output_path", os.path.join(tmp, "out"), "--stage_times"]
    for name, flag in (("train", "--train_vectors"), ("val", "--valid_vectors"), ("test", "--test_vectors")):
        argv += [flag, os.path.join(tmp, name + ".npy")]
    t0 = time.perf_counter()
    r = subprocess.run(argv, cwd=ROOT, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-3000:]); raise SystemExit(1)
    stages = json.loads([l for l in r.stderr.splitlines() if l.startswith('{"stage_seconds"')][-1])
    # the reference's own two lines on the train split's rank array (what write_neighbor_file replaces), on a sample
    from textreact_amd import neighbors as N
    got = json.load(open(os.path.join(tmp, "out", "train.json")))
    sample = min(n, 100000)
    rank = np.array([[int(x[2:]) for x in e["nn"]] for e in got[:sample]])
    ids = ["US%08d" % i for i in range(n)]
    t0 = time.perf_counter()
    res = [{'id': ids[i], 'nn': [ids[j] for j in nn]} for i, nn in enumerate(rank)]
    with open(os.path.join(tmp, "ref.json"), "w") as f:
        json.dump(res, f)
    t_ref = (time.perf_counter() - t0) * n / sample
    sizes = {name: os.path.getsize(os.path.join(tmp, "out", name + ".json")) for name in ("train", "val", "test")}
    print(json.dumps({"what": "python -m textreact_amd.retrieve_faiss --stage_times, %d x %d %s train vectors, 2 x %d queries, k = 20, L2" % (n, d, dt, nq),
                      "process_wall_s": round(wall, 2), "cli": stages, "json_bytes": sizes,
                      "reference_style_map_and_dump_train_split_s": round(t_ref, 2), "generate_inputs_s": round(t_gen, 1),
                      "self_is_first_neighbour": all(e["nn"][0] == e["id"] for e in got[:1000])}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
