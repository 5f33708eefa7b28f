"""Tanimoto brute force (retrieve/retrieve.py) on one MI355X: N train fingerprints of 2048 counts, Q queries, top 100.
One JSON line per stage: the scoring kernel against its VALU roofline (4 byte-differences per lane and v_sad_u8), the
whole search (scores + top-k + gather), and the CPU oracle on a bounded sample beside them.

    python bench_tanimoto.py [--n 680000] [--nq 128] [--k 100]
"""
import argparse
import json
import time

import numpy as np
import torch

from textreact_amd import tanimoto

VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9          # CUs x SIMDs x lanes x clock: one 32-bit VALU op per lane and cycle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=680000)      # USPTO_condition_train.csv is ~680 k reactions
    ap.add_argument("--nq", type=int, default=128)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--d", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    # difference-fingerprint-like counts: ~3 % of the positions occupied, magnitudes 1..4, signed
    def make(n):
        x = torch.randint(1, 5, (n, a.d), generator=g, device=dev, dtype=torch.int32)
        x = x * (torch.rand((n, a.d), generator=g, device=dev) < 0.03)
        return x * (1 - 2 * (torch.rand((n, a.d), generator=g, device=dev) < 0.5).int())
    corpus = torch.cat([make(min(100000, a.n - lo)) for lo in range(0, a.n, 100000)])
    queries = make(a.nq)
    idx = tanimoto.TanimotoIndex(a.d)
    t0 = time.perf_counter(); idx.add(corpus); torch.cuda.synchronize(); add_ms = (time.perf_counter() - t0) * 1e3

    q_t, q_sum = idx._pack_queries(queries)
    both = torch.empty((a.nq, a.n), dtype=torch.int16, device=dev)
    bmax = torch.empty((a.nq, (a.n + 63) // 64), dtype=torch.float32, device=dev)
    L = tanimoto.lib()

    def scores():
        tanimoto._check(L.trx_tanimoto_scores(idx.packed.data_ptr(), idx.row_sum.data_ptr(), a.n, a.d, q_t.data_ptr(), q_sum.data_ptr(),
                                              a.nq, both.data_ptr(), a.n, bmax.data_ptr(), tanimoto._stream(dev)))

    def timeit(fn, iters=10, warm=2):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    ms = timeit(scores)
    ms_plain = timeit(lambda: tanimoto._check(L.trx_tanimoto_scores(idx.packed.data_ptr(), idx.row_sum.data_ptr(), a.n, a.d, q_t.data_ptr(),
                                                                    q_sum.data_ptr(), a.nq, both.data_ptr(), a.n, None, tanimoto._stream(dev))))
    byte_ops = a.n * a.nq * a.d
    passes = -(-a.nq // 64)
    out = {"kernel": "tanimoto scores_kernel", "n": a.n, "d": a.d, "nq": a.nq, "ms": ms, "ms_without_block_maxima": ms_plain, "pairs_per_s": a.n * a.nq / (ms * 1e-3),
           "roofline": {"bound": "valu", "achieved": byte_ops / 4 / (ms * 1e-3) / 1e12, "peak": VALU_LANE_OPS / 1e12,
                        "unit": "T lane-ops/s (v_sad_u8: 4 counts each)", "frac": byte_ops / 4 / (ms * 1e-3) / VALU_LANE_OPS},
           "hbm": {"algorithmic_GB": (passes * a.n * a.d + a.nq * a.n * 2) / 1e9,
                   "achieved_GBps": (passes * a.n * a.d + a.nq * a.n * 2) / (ms * 1e-3) / 1e9}}
    print(json.dumps(out))
    ms_search = timeit(lambda: idx.search(queries, a.k), iters=5, warm=1)
    line = {"metric": "tanimoto top-%d queries/s over %dx%d fingerprints" % (a.k, a.n, a.d), "value": a.nq / (ms_search * 1e-3),
            "unit": "queries/s", "ms_per_batch": ms_search, "nq": a.nq, "add_ms": add_ms}
    if not a.no_cpu_baseline:
        from oracle import tanimoto as oracle            # the checker, timed beside the kernel (never part of the product path)
        c_host, q_host = corpus[:100000].cpu().numpy(), queries[:4].cpu().numpy()
        t0 = time.perf_counter(); oracle.search(q_host, c_host, a.k); dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": 4 / dt * (100000 / a.n), "unit": "queries/s", "cores": 1, "kind": "port",
                                "sample": "4 queries x 100000 of %d rows, numpy, scaled to the full corpus" % a.n}
        s_gpu, r_gpu = idx.search(queries[:4], a.k)
        idx2 = tanimoto.TanimotoIndex(a.d); idx2.add(corpus[:100000])
        s2, r2 = idx2.search(queries[:4], a.k)
        ws, wr = oracle.search(q_host, c_host, a.k)
        line["cpu_baseline"]["agreement_with_gpu"] = bool(np.array_equal(r2.cpu().numpy(), wr) and np.array_equal(s2.cpu().numpy(), ws))
    print(json.dumps(line))


if __name__ == "__main__":
    main()
