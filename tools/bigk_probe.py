"""The two-scan path (TRX_FAST_MAX_K < k <= TRX_WIDE_MAX_K) at size: a bf16 corpus of n x 768 rows, nq queries, Gaussian or
clustered (200 centres: the scores of a query fall off a cliff behind its own cluster, which the threshold guess cannot know),
several k.  Prints one JSON line per (data, metric, k): time, the tier counters, and a check of the first 32 queries against
fp64 torch.topk; and k = 1000 / 2048 (above TRX_WIDE_MAX_K: the exact scan for every query) for 256 queries.    python3 tools/bigk_probe.py [n [nq]]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import textreact_amd.faiss_compat as faiss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
d = 768
g = torch.Generator(device="cuda"); g.manual_seed(3)
for kind in ("gauss", "clustered"):
    if kind == "gauss":
        y = torch.randn((n, d), generator=g, device="cuda").bfloat16(); x = torch.randn((nq, d), generator=g, device="cuda").bfloat16()
    else:
        c = torch.randn((200, d), generator=g, device="cuda")
        y = (c[torch.randint(0, 200, (n,), generator=g, device="cuda")] + 0.3 * torch.randn((n, d), generator=g, device="cuda")).bfloat16()
        x = (c[torch.randint(0, 200, (nq,), generator=g, device="cuda")] + 0.3 * torch.randn((nq, d), generator=g, device="cuda")).bfloat16()
    for metric in ("IP", "L2"):
        idx = (faiss.IndexFlatIP if metric == "IP" else faiss.IndexFlatL2)(d)
        idx.add(y)
        s64 = x[:32].double() @ y.double().T
        if metric == "L2": s64 = 2 * s64 - (y.double() ** 2).sum(1)[None, :]
        for k in (24, 32, 100, 256, 1000, 2048):
            xs = x if k <= 256 else x[:256]
            idx.search(x[:512] if k <= 256 else x[:8], k)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            D, I = idx.search(xs, k)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            st = idx.last_stats()
            ref = s64.topk(k, dim=1).indices.sort(dim=1).values
            mine = torch.as_tensor(I[:32]).to(ref.device).sort(dim=1).values
            print(json.dumps({"what": "exact top-k, %d x %d bf16 corpus, %d queries, one MI355X" % (n, d, xs.shape[0]), "data": kind, "metric": metric, "k": k,
                              "search_ms": round((t1 - t0) * 1e3, 2), "queries_per_s": round(xs.shape[0] / (t1 - t0)),
                              "second_scan_unproven": st["n_rescored"], "third_scan": st["n_rescanned"], "exact_scan": st["n_uncertified"],
                              "id_sets_equal_fp64_topk_32q": bool((ref == mine).all().item())}), flush=True)
        del idx
