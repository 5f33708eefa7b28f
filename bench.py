#!/usr/bin/env python3
"""bench.py -- headline benchmark of the retrieval hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one exact top-10 search of ALL 65,536 queries against the 1,000,000 x 768 bf16 corpus
(BASELINE.json configs[1]).  At N > 1 (launched by torch.distributed.run, one rank per GPU) the
same corpus is row-sharded N ways (rank r holds rows [r*N/G, (r+1)*N/G)), queries are replicated,
per-shard (fp64 score, global id) lists are all-gathered over RCCL and merged on every rank:
strong scaling of the metric's fixed workload.  Inputs are synthetic (seeded Gaussian, rounded to
bf16) and resident in HBM before the timed region.  `value` = queries / second of the whole job.

Also reported on the same JSON line:
  roofline     -- the scan kernel (knn_scan_kernel) against the dense bf16 MFMA peak: algorithmic
                  FLOPs 2*Q*N_local*d per launch / mean launch duration from HIP events recorded
                  around the launch on its own stream inside libtrxknn.so (trx_search_stats.scan_ms);
                  hbm_frac = the algorithmic bytes 2(N d + Q d) + 8 Q k of the same launch against
                  8 TB/s ("report both fractions"; the contraction is not HBM-bound);
                  traffic = L2-miss bytes per launch from the committed PMC passes, with the
                  commit they were taken at (traffic_source)
  ms_per_step  -- wall clock over the K timed steps (the contract); ms_per_step_median = median of
                  >= 10 steps each timed with HIP events on the stream the kernels run on;
                  ms_per_step_with_d2h = the same step plus the copy of D and I to pinned host memory
  cpu_baseline -- the oracle's FAISS restatement (host BLAS sgemm blocks + heap) timed on this
                  host's cores on a bounded query sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CORPUS, DIM, N_QUERIES, TOPK = 1_000_000, 768, 65_536, 10
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)


def make_rows(n, d, seed, device, row0=0):
    """Rows [row0, row0+n) of the seeded Gaussian matrix, bf16, generated on the device in
    fixed 65,536-row blocks so that any sharding sees the same values."""
    import torch
    blk = 65536
    out = torch.empty((n, d), dtype=torch.bfloat16, device=device)
    b0 = row0 // blk
    pos = 0
    while pos < n:
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + b0)
        full = torch.randn((blk, d), generator=g, device=device, dtype=torch.float32).to(torch.bfloat16)
        lo = (row0 + pos) - b0 * blk
        take = min(blk - lo, n - pos)
        out[pos:pos + take] = full[lo:lo + take]
        pos += take
        b0 += 1
    return out


def try_faiss():
    """The reference's own library (retrieve/retrieve_faiss.py:14).  It is not in the build image; a box that has it
    makes it the CPU baseline and the parity reference."""
    try:
        import faiss
        return faiss, str(getattr(faiss, "__version__", "unknown"))
    except Exception:
        return None, "unavailable"


def cpu_baseline(corpus_dev, queries_dev, k, target_seconds=20.0, metric=0):
    """CPU search on a bounded sample of the same workload: real FAISS when importable ('reference'), otherwise the
    oracle's FAISS-structured restatement ('port': fp32 sgemm of 4096 x 1024 blocks by the host BLAS + the strict-
    admission heap on every block).  ONE full FAISS query block (4,096 queries) against the whole corpus
    unless a probe predicts more than 120 s.  The port spreads the corpus blocks over one worker thread per core with a
    single BLAS thread per sgemm call (oracle.knn_faiss_blas_mt); FAISS's own form -- all threads inside each 6.4-GFLOP
    sgemm -- is probed too and its rate reported beside it (`blas_internal_threading_tflops`)."""
    import numpy as np
    from oracle import flat_knn as oracle
    cores = len(os.sched_getaffinity(0))
    y = corpus_dev.float().cpu().numpy()
    d = y.shape[1]
    faiss, faiss_version = try_faiss()
    if faiss is not None:
        ref = (faiss.IndexFlatL2 if metric == 1 else faiss.IndexFlatIP)(d)
        ref.add(y)
        nq = min(4096, queries_dev.shape[0])
        x = queries_dev[:nq].float().cpu().numpy()
        ref.search(x[:64], k)
        t0 = time.perf_counter(); D, I = ref.search(x, k); t1 = time.perf_counter()
        return {"value": nq / (t1 - t0), "unit": "queries/s", "cores": cores, "kind": "reference", "faiss": faiss_version,
                "tflops": 2.0 * nq * y.shape[0] * d / (t1 - t0) / 1e12,
                "sample": "%d of %d queries x full %dx%d corpus, faiss.IndexFlat%s %s" % (
                    nq, queries_dev.shape[0], y.shape[0], d, "L2" if metric == 1 else "IP", faiss_version)}, I
    blas = "unknown"
    try:
        from threadpoolctl import threadpool_info
        blas = "; ".join("%s %s (%s, %d threads max)" % (i.get("internal_api"), i.get("version"), i.get("threading_layer"), i.get("num_threads"))
                         for i in threadpool_info() if i.get("user_api") == "blas") or "unknown"
    except Exception:
        pass
    nprobe = min(4096, queries_dev.shape[0])
    probe = queries_dev[:nprobe].float().cpu().numpy()
    # (a) FAISS's own threading: every BLAS thread inside one block's sgemm.  8 corpus blocks.
    ysub = y[:8192]
    oracle.knn_faiss_blas(metric, probe[:64], ysub[:1024], k)  # warm BLAS threads
    t0 = time.perf_counter(); oracle.knn_faiss_blas(metric, probe, ysub, k); t1 = time.perf_counter()
    internal_tflops = 2.0 * nprobe * ysub.shape[0] * d / (t1 - t0) / 1e12
    # (b) one worker per core over the corpus blocks, one BLAS thread each: probe on 2 blocks per worker, then the sample.
    # Two candidates for the block sgemm, the faster one on the probe runs the sample: torch's (MKL, the BLAS conda's
    # faiss-cpu links) from one worker per core, and numpy's (OpenBLAS) from at most 64 workers -- its build carries a table
    # for 64 calling threads, and 256 callers overran it (a warning, "Bad memory unallocation", a crash at exit).
    import torch
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    mkl = lambda a, b: torch.mm(torch.from_numpy(a), torch.from_numpy(b).t()).numpy()      # noqa: E731
    cands = [("torch.mm (MKL)", cores, mkl), ("numpy (OpenBLAS)", min(cores, 64), None)]
    try:
        best, probes = None, {}
        for name, workers, gemm in cands:
            yprobe = y[:min(y.shape[0], 2 * 1024 * workers)]
            oracle.knn_faiss_blas_mt(metric, probe[:256], yprobe[:1024 * min(workers, 8)], k, workers, gemm=gemm)
            t0 = time.perf_counter(); oracle.knn_faiss_blas_mt(metric, probe, yprobe, k, workers, gemm=gemm); t1 = time.perf_counter()
            rate = nprobe * yprobe.shape[0] / (t1 - t0)
            probes[name] = {"workers": workers, "tflops": 2.0 * rate * d / 1e12, "corpus_rows_in_probe": int(yprobe.shape[0])}
            if best is None or rate > best[0]:
                best = (rate, name, workers, gemm)
        rate, blas_name, workers, gemm = best
        est_full = nprobe * y.shape[0] / rate
        nq = nprobe if est_full <= 120.0 else max(512, int(nprobe * target_seconds / est_full))
        x = probe[:nq]
        t0 = time.perf_counter(); D, I = oracle.knn_faiss_blas_mt(metric, x, y, k, workers, gemm=gemm); t1 = time.perf_counter()
    finally:
        torch.set_num_threads(torch_threads)
    blas_mt = blas_name + (": " + next((l.strip(" -") for l in torch.__config__.show().splitlines() if "Math Kernel" in l), "")[:70] if gemm is not None else ": " + blas)
    return {"value": nq / (t1 - t0), "unit": "queries/s", "cores": workers, "host_cores": cores, "kind": "port", "faiss": faiss_version,
            "tflops": 2.0 * nq * y.shape[0] * d / (t1 - t0) / 1e12, "blas": blas_mt, "blas_of_the_internal_threading_probe": blas,
            "threads": "%d worker threads over the corpus blocks x 1 BLAS thread per sgemm call" % workers,
            "blas_internal_threading_tflops": internal_tflops, "seconds": t1 - t0,
            # why the candidate that ran is the one that ran: both candidates' measured rates on the same probe (the faster takes the sample)
            "candidates_on_the_probe": probes,
            "sample": "%d of %d queries (%s) x full %dx%d corpus, fp32 host-BLAS sgemm 4096x1024 blocks + heap"
                      % (nq, queries_dev.shape[0], "one full FAISS query block" if nq == 4096 else "probe predicted %.0f s for 4096" % est_full,
                         y.shape[0], d)}, I


def selfcheck(index, local, shard, lo, queries, k, D, I, world, rank, dev, n_sample=32, row_groups=None):
    """Before the timed region: (1) every rank holds the same (D, I) -- a hash per rank, gathered; (2) `n_sample` queries
    are re-done with nothing of the search path in it: fp64 scores of the queries against this rank's shard by torch.matmul,
    a local top-k, ONE plain all_gather of the (score, global id) lists, a sort by (score desc, id asc) on every rank.  A
    wrong exchange (split sizes, offsets, the merge's list order, a stale buffer) shows up as a non-zero exit code, not as a
    fast wrong number.  Returns the list of problems (empty = fine)."""
    import torch
    import torch.distributed as dist
    problems = []
    h = torch.stack([(I * torch.arange(1, k + 1, device=dev)).sum(), I.sum(), D.view(torch.int32).long().sum()])
    if world > 1:
        hs = [torch.empty_like(h) for _ in range(world)]
        if dist.get_backend() == "gloo":
            hh = [x.cpu() for x in hs]; dist.all_gather(hh, h.cpu()); hs = hh
        else:
            dist.all_gather(hs, h)
        if any(not torch.equal(x.cpu(), hs[0].cpu()) for x in hs):
            problems.append("ranks hold different (D, I): %r" % [x.tolist() for x in hs])
    g = torch.Generator().manual_seed(11)
    rows = torch.randperm(queries.shape[0], generator=g)[:n_sample].to(dev)
    sc = queries[rows].double() @ shard.double().t()                       # [n_sample, n_local] fp64
    kk = min(k, sc.shape[1])
    top, idx = torch.topk(sc, kk, dim=1)
    ids = idx + lo
    if world > 1:
        pack = torch.stack([top.view(torch.int64), ids], dim=0).contiguous()
        parts = [torch.empty_like(pack) for _ in range(world)]
        if dist.get_backend() == "gloo":
            hp = [x.cpu() for x in parts]; dist.all_gather(hp, pack.cpu()); parts = [x.to(dev) for x in hp]
        else:
            dist.all_gather(parts, pack)
        parts = parts[:row_groups or world]      # a rows x queries grid holds every row shard once per column: column 0 = ranks 0 .. Gr - 1
        top = torch.cat([p_[0].view(torch.float64) for p_ in parts], dim=1)
        ids = torch.cat([p_[1] for p_ in parts], dim=1)
    order = torch.argsort(ids, dim=1, stable=True)                        # (score desc, id asc): sort by id, then stably by score
    top, ids = torch.gather(top, 1, order), torch.gather(ids, 1, order)
    order = torch.argsort(top, dim=1, descending=True, stable=True)[:, :k]
    want_s, want_i = torch.gather(top, 1, order), torch.gather(ids, 1, order)
    got_i, got_d = I[rows], D[rows]
    # the matmul's summation order differs from the product's k-ordered fma chain, so two rows whose fp64 scores agree to the
    # last bits may come out swapped: compare the neighbour SETS of a query (a swap inside the top k changes nothing there);
    # a mislabelled id, a lost shard or a wrong merge changes the set
    if not torch.equal(torch.sort(got_i, dim=1).values, torch.sort(want_i, dim=1).values):
        nbad = int((torch.sort(got_i, dim=1).values != torch.sort(want_i, dim=1).values).any(dim=1).sum())
        problems.append("rank %d: the neighbours of %d of %d sampled queries differ from the independent fp64 re-computation" % (rank, nbad, len(rows)))
    if not bool(torch.allclose(got_d.double(), want_s, rtol=1e-6, atol=1e-6)):
        problems.append("rank %d: distances of the sampled queries differ from the fp64 re-computation" % rank)
    return problems


def fingerprint_workload(args, dev, local_rank):
    """retrieve/retrieve_faiss.py:62-74 as the reference runs it: IndexFlatL2, k = 20, d = 2048 sparse signed counts
    (reaction difference fingerprints, :18-27), the training set searching itself (:114-115).  One step = the whole
    self-search (680,000 queries in batches of 65,536 = 11 scan launches).  Not the headline metric: a second line with
    its own roofline (same kernel, K = 2048 -> 32 K-steps per tile) and CPU baseline, recorded under profiles/.
    --workload morgan: the molecule form of the same call (retrieve/retro.sh: --field product_smiles -> :36-44, Morgan bit
    vectors of 1024 components, 0 or 1, 5 % set), which the scan runs on fp4 operands."""
    import torch
    import textreact_amd.faiss_compat as faiss
    n = args.n_corpus if args.n_corpus != N_CORPUS else 680_000     # ~ USPTO-condition train size
    g = torch.Generator(device=dev); g.manual_seed(1)
    morgan = args.workload == "morgan"
    dim = 1024 if morgan else 2048
    y = torch.empty((n, dim), dtype=torch.bfloat16, device=dev)
    for r0 in range(0, n, 65536):                                   # blockwise: the int64 temporaries of the whole set are 11 GB
        m = min(65536, n - r0)
        mask = torch.rand((m, dim), generator=g, device=dev) < (0.05 if morgan else 0.02)
        vals = torch.ones((m, dim), dtype=torch.int64, device=dev) if morgan else torch.randint(-10, 11, (m, dim), generator=g, device=dev)
        y[r0:r0 + m] = (mask * vals).to(torch.bfloat16)
    del mask, vals
    idx = faiss.IndexFlatL2(dim, device=local_rank)
    idx.set_timing(True)
    idx.add(y)
    for _ in range(args.warmup):
        idx.search(y[:65536], 20)
    torch.cuda.synchronize()
    scan_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(max(1, min(args.steps, 3))):
        D, I = idx.search(y, 20)
        st = idx.last_stats()
        scan_ms += st["scan_ms"]; launches += st["scan_launches"]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steps = max(1, min(args.steps, 3))
    ok = bool((I[:, 0] == torch.arange(n, device=dev)).float().mean() > 0.99) and bool((D[:, 0] == 0).all())
    flops_step = 2.0 * n * n * dim
    form = st.get("int8_scan", 0)
    peak = {0: PEAK_BF16_TFLOPS, 1: 5000.0, 2: 10000.0}[form]
    mean_launch_ms = scan_ms / max(launches, 1)
    achieved = flops_step * steps / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    # The second roofline of these forms: the L2 -> LDS fill.  A K-step of a workgroup stages 32 KiB of corpus rows and 32 KiB
    # of query rows (128 bytes per row) whatever the element type, so with the MFMA time of a K-step halved (int8) or quartered
    # (fp4) and its fill bytes unchanged the loop runs into the rate at which the LDS-DMA path delivers: 14.0 TB/s over the chip
    # in this access pattern (tools/fill_bench.hip, DESIGN_HISTORY 3.1).  Per step: query tiles x corpus tiles x K-steps x 64 KiB.
    row_bytes = {0: 2 * st["k_split"], 1: st["k_split"], 2: st["k_split"] // 2}[form]
    fill_step = float(-(-n // 256)) * float(-(-n // 256)) * (row_bytes // 128) * 65536.0
    fill_tbps = fill_step * steps / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    traffic, traffic_source = forms_traffic("morgan" if morgan else "fingerprint", n, form)
    line = {"metric": "queries/sec, train self-search IndexFlatL2 k=20 over %dx%d integer fingerprints (the reference's own workload)" % (n, dim),
            "value": n * steps / (t1 - t0), "unit": "queries/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
            "ms_per_step": (t1 - t0) / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("bf16", "i8", "fp4")[form], "data": "synthetic",
            "config": {"workload": ("exact L2 top-20, %dx1024 Morgan-like bit vectors (5 %% set), the set searching itself" if morgan else
                                    "exact L2 top-20, %dx2048 sparse signed-count fingerprints (2 %% dense, |v| <= 10), the set searching itself")
                                   % n, "corpus_rows": n, "dim": dim, "queries": n, "k": 20, "exact_class": st["exact_class"],
                       "int8_scan": form,      # 1: the scan's int8 form (v_mfma_i32_16x16x64_i8), 2: its fp4 form (v_mfma_f32_16x16x128_f8f6f4); TRX_NO_FP4=1 / TRX_NO_I8=1 for the others
                       "uncertified_queries_per_step": st["n_uncertified"], "self_is_first": ok,
                       "scan_launches_per_step": launches // steps},
            "roofline": {"bound": "mfma", "kernel": "knn_scan_kernel", "achieved": achieved, "peak": peak,
                         "unit": ("TFLOP/s", "Top/s (dense int8 MFMA peak)", "TFLOP/s (dense fp4 MFMA peak: twice fp8's)")[form],
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source, "launch_ms": mean_launch_ms,
                         "flops_per_launch": flops_step * steps / max(launches, 1),
                         "fill": {"bound": "L2 -> LDS fill (LDS-DMA)", "achieved": fill_tbps, "peak": 14.0, "unit": "TB/s", "frac": fill_tbps / 14.0,
                                  "bytes_per_step": fill_step, "peak_source": "tools/fill_bench.hip: this loop's access pattern, LDS-DMA, measured"}}}
    if args.host_api:
        line.update(host_api_leg(idx, y, I, "int8" if morgan else "int64", local_rank))
    if not args.no_cpu_baseline:
        base, I_cpu = cpu_baseline(y, y, 20, metric=1)
        import numpy as np
        base["index_agreement_with_gpu"] = float((I[: I_cpu.shape[0]].cpu().numpy() == I_cpu).mean())
        line["cpu_baseline"] = base
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line))


def dense_host_api_leg(shard, queries, k, I_dev, local_rank):
    """The drop-in call itself on the headline workload, beside `value` (never instead of it): the FAISS protocol as a dense
    retriever's script makes it (retrieve_faiss.py:62-74 with IndexFlatIP) -- `index.add(corpus)` and `index.search(queries, k)` on
    HOST float32 NumPy arrays, host (D, I) back.  PCIe both ways and the library's staging are inside `value_host_api`."""
    import numpy as np
    import textreact_amd.faiss_compat as faiss
    from textreact_amd import _lib
    yh = shard.float().cpu().numpy(); xh = queries.float().cpu().numpy()
    idx = faiss.IndexFlatIP(yh.shape[1], device=local_rank)
    idx.add(yh[:4096]); idx.search(xh[:4096], k); idx.reset()            # warm: worker pool, pinned buffers
    t0 = time.perf_counter(); idx.add(yh); t_add = time.perf_counter() - t0
    best, Ih = None, None
    for _ in range(3):
        t0 = time.perf_counter(); Dh, Ih = idx.search(xh, k); t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    same = bool(np.array_equal(Ih, I_dev.cpu().numpy()))
    return {"value_host_api": xh.shape[0] / best,
            "host_api": {"what": "IndexFlatIP.add / .search on host numpy float32 arrays (the FAISS protocol), host (D, I) out: PCIe and staging included; "
                                 "not `value`", "add_ms": t_add * 1e3, "search_ms": best * 1e3, "corpus_GB": yh.nbytes / 1e9, "queries_MB": xh.nbytes / 1e6,
                         "host_threads": _lib.lib().trx_host_threads(), "same_ids_as_device_resident": same}}


def forms_traffic(workload, n, form, root=ROOT):
    """roofline.traffic of the int8 / fp4 forms: L2-miss bytes per scan launch from profiles/traffic_forms.json (profiles/pmc_forms.sh:
    one --pmc pass per counter), reported only for the size it was taken at and the kernel text this run compiled from"""
    try:
        tj = json.load(open(os.path.join(root, "profiles", "traffic_forms.json")))
    except Exception:
        return None, None
    if tj.get("scan_source_sha256") != scan_source_sha256(root):
        return None, "profiles/traffic_forms.json is STALE for this knn_scan.hip / knn_common.h (source hash differs): traffic not reported"
    if n != {"fingerprint": 680_000, "morgan": 800_000}[workload]:
        return None, "profiles/traffic_forms.json was taken at another corpus size"
    if form != {"fingerprint": 1, "morgan": 2}[workload]:
        return None, "profiles/traffic_forms.json holds this workload's default form (int8 / fp4), not the one this run was switched to"
    e = tj.get(workload) or {}
    return e.get("hbm_bytes_per_launch"), "profiles/traffic_forms.json@%s (same scan-kernel source; kernel trace %.2f ms per launch there)" % (
        tj.get("tag"), e.get("avg_launch_ms_kernel_trace") or float("nan"))


def host_api_leg(idx_dev, y, I_dev, np_dtype, local_rank):
    """--host-api: the same self-search as the reference's CLI makes it (retrieve_faiss.py:62-74) -- HOST numpy arrays of the
    dtype RDKit's fingerprints arrive in (int64 counts :24-27, int8 bits :36-44) through index.add / index.search, host (D, I)
    back.  PCIe and the library's narrowing pass are inside this figure; it is a second number, never `value`."""
    import numpy as np
    import torch
    import textreact_amd.faiss_compat as faiss
    from textreact_amd import _lib
    n, dim = y.shape
    xh = np.empty((n, dim), dtype=np_dtype)
    for r0 in range(0, n, 65536):
        xh[r0:r0 + 65536] = y[r0:r0 + 65536].float().cpu().numpy().astype(np_dtype)
    idx = faiss.IndexFlatL2(dim, device=local_rank)
    idx.add(xh[:4096]); idx.search(xh[:4096], 20); idx.reset()      # warm: worker pool, pinned buffers
    t0 = time.perf_counter(); idx.add(xh); t_add = time.perf_counter() - t0
    idx.search(xh[:70000], 20)
    best, Ih = None, None
    for _ in range(2):
        t0 = time.perf_counter(); Dh, Ih = idx.search(xh, 20); t = time.perf_counter() - t0
        best = t if best is None else min(best, t)
    same = bool(np.array_equal(Ih, I_dev.cpu().numpy()))
    return {"value_host_api": n / best, "host_api": {"what": "index.add / index.search on host numpy %s arrays, host (D, I) out: PCIe and the "
                                                              "library's narrowing pass included" % np_dtype,
                                                     "add_ms": t_add * 1e3, "search_ms": best * 1e3, "input_GB": xh.nbytes / 1e9,
                                                     "host_threads": _lib.lib().trx_host_threads(), "host_cores": len(os.sched_getaffinity(0)),
                                                     "same_ids_as_device_resident": same, "int8_scan": idx.last_stats()["int8_scan"]}}


SCAN_SOURCES = ("textreact_amd/csrc/knn_scan.hip", "textreact_amd/csrc/knn_common.h")


def scan_source_sha256(root=ROOT):
    """sha256 over the sources the scan kernel is compiled from.  profiles/summarize.py stores it beside the PMC figures
    (the GPU box has no .git, so a commit id cannot be checked there; the text of the kernel can)."""
    import hashlib
    h = hashlib.sha256()
    for rel in SCAN_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def committed_traffic(root=ROOT):
    """roofline.traffic: L2-miss bytes per scan launch from the separate --pmc passes of profiles/run_profile.sh, a committed
    constant -- reported ONLY when it was measured on the kernel text this run compiled from; otherwise null, and the
    source field says why."""
    tf = os.path.join(root, "profiles", "traffic.json")
    try:
        tj = json.load(open(tf))
    except Exception:
        return None, None
    want = tj.get("scan_source_sha256")
    if want is None or want != scan_source_sha256(root):
        return None, "profiles/traffic.json@%s is STALE for this knn_scan.hip / knn_common.h (source hash differs): traffic not reported" % tj.get("git_sha", "unknown")
    return tj.get("hbm_bytes_per_launch"), "profiles/traffic.json@%s (same scan-kernel source as this run; kernel trace %.2f ms per launch there)" % (
        tj.get("git_sha", "unknown"), tj.get("avg_launch_ms_kernel_trace") or float("nan"))


def launch_or_refuse(args):
    """--gpus N must be what runs.  Decided before anything of this process touches the GPU (no torch import yet):
      * WORLD_SIZE set (a launcher started us) and != --gpus: exit 2 -- a line that says n_gpus = WORLD_SIZE under a command
        that says --gpus N is a mis-measurement, not a warning;
      * WORLD_SIZE unset and --gpus N > 1: this process becomes the launcher's parent -- it starts
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
        arguments>` as a CHILD process (never exec), relays its output (rank 0's JSON line) and exits with its code."""
    from textreact_amd import _dist      # (imports nothing that touches the GPU)
    _dist.refuse_mismatch(args.gpus, "bench.py")
    if os.environ.get("WORLD_SIZE") is not None or args.gpus <= 1:
        return
    sys.exit(_dist.launch_children(args.gpus, __file__, sys.argv[1:], port_env="TRX_BENCH_MASTER_PORT"))      # the child inherits stdout / stderr: rank 0's JSON line goes straight through


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-corpus", type=int, default=N_CORPUS)
    ap.add_argument("--n-queries", type=int, default=N_QUERIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: --n-corpus rows PER GPU (BASELINE.json configs[2]: 8 GPUs x 1M rows = 8M); "
                         "default is strong scaling, the 1M-row corpus row-sharded over the GPUs")
    ap.add_argument("--replicas", action="store_true",
                    help="query-sharded replicas (SURVEY 8e, the separate line): every GPU holds the WHOLE corpus and searches "
                         "its 1/G of the queries, no collective on the data path; not the north-star line (row-sharded)")
    ap.add_argument("--row-groups", type=int, default=0,
                    help="rows x queries grid (sharded.ShardedFlatIndex(row_groups=Gr)): the corpus row-sharded Gr ways, the queries split "
                         "over the G / Gr columns; 0 = pure row sharding (Gr = G), the north-star line")
    ap.add_argument("--finish-first", action="store_true",
                    help="row-sharded search: finish the local search (one host wait) before the exchange is enqueued, instead of enqueuing "
                         "it behind the search and agreeing on late fall-backs with a one-word all-reduce (ShardedFlatIndex(stream_ordered=False))")
    ap.add_argument("--selfcheck", action="store_true",
                    help="also at N = 1 (always on at N > 1): before the timed region, compare every rank's result hashes and re-do "
                         "32 sampled queries by an independent fp64 matmul + all_gather + sort; exit code 3 on a mismatch")
    ap.add_argument("--host-api", action="store_true",
                    help="--workload fingerprint / morgan: also run the self-search through the HOST protocol the reference calls "
                         "(numpy int64 / int8 arrays in, numpy (D, I) out) and report it as value_host_api")
    ap.add_argument("--workload", default="dense", choices=["dense", "fingerprint", "morgan"],
                    help="dense = the headline (BASELINE.json configs[1]); fingerprint = the reference's own call: "
                         "IndexFlatL2, k=20, 2048-d integer reaction fingerprints, train set searching itself (not the headline)")
    args = ap.parse_args()
    launch_or_refuse(args)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the retrieval path has no CPU fallback")
    # TRX_BENCH_BACKEND=gloo TRX_BENCH_DEVICE=0 lets several ranks share one GPU to rehearse the N > 1 path
    # (RCCL refuses two ranks on one device); the real run is nccl = RCCL, one GPU per rank.  The group itself is set up where
    # every N > 1 entry point sets it up: textreact_amd/_dist.py (set_device, then init_process_group("nccl", device_id=...))
    from textreact_amd import _dist
    for mine, theirs in (("TRX_BENCH_BACKEND", "TRX_DIST_BACKEND"), ("TRX_BENCH_DEVICE", "TRX_DEVICE")):
        if mine in os.environ:
            os.environ[theirs] = os.environ[mine]
    backend = os.environ.get("TRX_DIST_BACKEND", "nccl")
    rank, world, dev = _dist.setup()
    local_rank = dev.index

    import textreact_amd.faiss_compat as faiss
    from textreact_amd.sharded import ShardedFlatIndex, shard_bounds

    n, d, nq, k = args.n_corpus * (world if args.weak else 1), DIM, args.n_queries, TOPK
    if args.workload in ("fingerprint", "morgan"):
        return fingerprint_workload(args, dev, local_rank)
    row_groups = args.row_groups if (args.row_groups and not args.replicas) else world
    if world % row_groups:
        raise SystemExit("bench.py: --row-groups %d does not divide %d ranks" % (row_groups, world))
    lo, hi = (0, n) if args.replicas else shard_bounds(n, row_groups, rank % row_groups)
    if args.weak and not args.replicas:      # "8 shards x 1,000,000, shard s uses seed 1234 + s"
        if row_groups != world:
            raise SystemExit("bench.py: --weak is the pure row-sharded line (one seed per rank); not with --row-groups")
        shard = make_rows(hi - lo, d, 1234 + rank, dev)
    else:
        shard = make_rows(hi - lo, d, 1234, dev, row0=lo)
    queries = make_rows(nq, d, 5678, dev)
    if args.replicas:
        qlo, qhi = shard_bounds(nq, world, rank)
        queries = queries[qlo:qhi].contiguous()
    local = faiss.IndexFlatIP(d, device=local_rank)
    local.set_timing(True)
    if args.replicas:
        index = local                        # a plain flat index per GPU: nothing to exchange
        index.add(shard)
    else:
        index = ShardedFlatIndex(d, faiss.METRIC_INNER_PRODUCT, local_index=local, row_groups=row_groups, stream_ordered=not args.finish_first)
        # fault injection for tests/test_bench_gpu.py: the last rank reports its rows one id too high -- the kind of
        # plumbing error the selfcheck exists for
        wrong = 1 if (os.environ.get("TRX_BENCH_INJECT_FAULT") == "offset" and world > 1 and rank == world - 1) else 0
        index.add_shard(shard, lo + wrong, n)
    torch.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        D, I = index.search(queries, k)
    checked = None
    if (world > 1 or args.selfcheck) and not args.replicas:
        D, I = index.search(queries, k)
        problems = selfcheck(index, local, shard, lo, queries, k, D, I, world, rank, dev, row_groups=row_groups)
        flag = torch.tensor([len(problems)], dtype=torch.int32, device=dev)
        if world > 1:
            if backend == "nccl":
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            else:
                fh = flag.cpu(); dist.all_reduce(fh, op=dist.ReduceOp.MAX); flag = fh
        if problems:
            print("bench.py selfcheck FAILED on rank %d: %s" % (rank, "; ".join(problems)), file=sys.stderr, flush=True)
        if int(flag.item()):
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
        checked = "passed: identical (D, I) hashes on %d rank(s); 32 sampled queries equal an independent fp64 matmul + all_gather + sort" % world
    scan_ms, launches, uncert = 0.0, 0, 0
    # per-step HIP events on torch's current stream = the stream the library launches on (faiss_compat passes it through
    # the C ABI): recording them costs nothing inside the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(args.steps, 10))]
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        D, I = index.search(queries, k)
        ev[i][1].record()
        st = local.last_stats()
        scan_ms += st["scan_ms"]; launches += st["scan_launches"]; uncert += st["n_uncertified"]
    sync()
    t1 = time.perf_counter()
    # "median of >= 10 iterations, hipEvent timing; a second number includes D2H of I".  Outside the region
    # the contract times: more event-timed steps up to 10 samples, then three steps that also copy D and I to the host.
    for i in range(args.steps, len(ev)):
        ev[i][0].record()
        index.search(queries, k)
        ev[i][1].record()
    sync()
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    D_host = torch.empty(D.shape, dtype=D.dtype).pin_memory()
    I_host = torch.empty(I.shape, dtype=I.dtype).pin_memory()
    d2h_ms = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        Dd, Id = index.search(queries, k)
        D_host.copy_(Dd, non_blocking=True); I_host.copy_(Id, non_blocking=True)
        b.record(); b.synchronize()
        d2h_ms.append(a.elapsed_time(b))
    sync()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = nq * args.steps / elapsed
        # algorithmic: 2*Q_local*N_local*d per step, over the scan launches the step took (one per 65,536 queries)
        q_rank = shard_bounds(queries.shape[0], world // row_groups, rank // row_groups) if not args.replicas else (0, queries.shape[0])
        q_rank = q_rank[1] - q_rank[0]           # the queries THIS rank searches (all of them unless the grid splits them)
        flops_launch = 2.0 * q_rank * (hi - lo) * d * args.steps / max(launches, 1)
        mean_launch_ms = scan_ms / max(launches, 1)
        achieved = flops_launch / (mean_launch_ms * 1e-3) / 1e12 if mean_launch_ms > 0 else 0.0
        q_launch = q_rank * args.steps / max(launches, 1)
        alg_bytes = 2.0 * ((hi - lo) * d + q_launch * d) + 8.0 * q_launch * k      # SURVEY 8d, per launch
        hbm_gbs = alg_bytes / (mean_launch_ms * 1e-3) / 1e9 if mean_launch_ms > 0 else 0.0
        traffic, traffic_source = None, None
        if world == 1 and n == N_CORPUS and nq == N_QUERIES:   # measured for exactly this launch
            traffic, traffic_source = committed_traffic()
        line = {
            "metric": "queries/sec top-10 over 1Mx768 corpus", "value": value, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "ms_per_step_median": step_ms[len(step_ms) // 2], "ms_per_step_min": step_ms[0], "timed_steps_for_median": len(step_ms),
            "ms_per_step_with_d2h": sorted(d2h_ms)[1],
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": ("exact IP top-%d, %dx%d bf16 corpus replicated on %d GPUs, %d queries per step split over them"
                                    if args.replicas else
                                    "exact IP top-%d, %dx%d bf16 corpus row-sharded %d-way, %d queries per step") % (k, n, d, world, nq),
                       "corpus_rows": n, "dim": d, "queries": nq, "k": k,
                       "parallelism": ("query-sharded replicas x%d (whole corpus per GPU, no collective)" % world if args.replicas else
                                       ("corpus row-sharded x%d + RCCL all-gather merge" % world) if row_groups == world else
                                       ("rows x queries grid %d x %d: corpus row-sharded x%d, queries split over %d columns, all-to-all inside a column + all-gather"
                                        % (row_groups, world // row_groups, row_groups, world // row_groups))) if world > 1 else "single GPU",
                       "transport": ("RCCL (nccl backend), one GPU per rank" if backend == "nccl" else
                                     "REHEARSAL: %d ranks share GPU %d, %s backend (host round trip in the all-gather); not a scaling measurement"
                                     % (world, local_rank, backend)) if world > 1 else None,
                       "exchange": (None if world == 1 or args.replicas else "finish first, then the exchange" if args.finish_first else
                                    "stream-ordered behind the local search + a one-word agreement all-reduce"),
                       "uncertified_queries_per_step": uncert / args.steps, "selfcheck": checked},
            "roofline": {"bound": "mfma", "kernel": "knn_scan_kernel", "achieved": achieved,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source, "launch_ms": mean_launch_ms,
                         "flops_per_launch": flops_launch,
                         "hbm_frac": hbm_gbs / 8000.0, "hbm_achieved_GBps": hbm_gbs, "hbm_peak_GBps": 8000.0,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # the rate the loop's operands reach the LDS at, against what the LDS-DMA path delivers in this access pattern
                         # (tools/fill_bench.hip: 14.0 TB/s): query tiles x corpus tiles x K-steps x 64 KiB per launch
                         "fill": (lambda fb: {"bound": "L2 -> LDS fill (LDS-DMA)", "achieved": fb / (mean_launch_ms * 1e-3) / 1e12 if mean_launch_ms > 0 else 0.0,
                                              "peak": 14.0, "unit": "TB/s", "frac": (fb / (mean_launch_ms * 1e-3) / 1e12 / 14.0) if mean_launch_ms > 0 else 0.0,
                                              "bytes_per_launch": fb, "peak_source": "tools/fill_bench.hip: this loop's access pattern, LDS-DMA, measured"})(
                             float(-(-int(q_launch) // 256)) * float(-(-(hi - lo) // 256)) * (2 * st["k_split"] // 128) * 65536.0)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line.update(dense_host_api_leg(shard, queries, k, I, local_rank))
            base, I_cpu = cpu_baseline(shard, queries, k)
            # the sample doubles as an end-to-end check: same neighbours as the GPU result
            # (fp32 BLAS order differs from the canonical fp64 order only on near-ties)
            import numpy as np
            agree = float((I[: I_cpu.shape[0]].cpu().numpy() == I_cpu).mean())
            base["index_agreement_with_gpu"] = agree
            line["cpu_baseline"] = base
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
