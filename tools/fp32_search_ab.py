"""fp32 inputs (what FAISS users hold: the external dense retriever's embeddings, README.md:44-47 of the reference) through the
index's two forms for data that bf16 does not hold: the approx mode (round 4: bf16 rounding, K = d, listing slack + wide re-score)
and the three-term split (TRX_FP32_SPLIT=1: K = 3d).  One form per process: python3 tools/fp32_search_ab.py [n [nq]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import textreact_amd.faiss_compat as faiss
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
g = torch.Generator(device="cuda"); g.manual_seed(1)
y = torch.randn((n, 768), generator=g, device="cuda"); x = torch.randn((nq, 768), generator=g, device="cuda")
for metric in ("IP", "L2"):
    idx = (faiss.IndexFlatIP if metric == "IP" else faiss.IndexFlatL2)(768)
    idx.set_timing(True)
    idx.add(y)
    idx.search(x[:4096], 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    D, I = idx.search(x, 10)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = idx.last_stats()
    import json
    print(json.dumps({"what": "exact top-10, %d x 768 fp32 corpus, %d fp32 queries, one MI355X" % (n, nq), "metric": metric,
                      "form": "three-term split (TRX_FP32_SPLIT=1)" if os.environ.get("TRX_FP32_SPLIT") else
                              ("approx mode, a-priori rounding bound (TRX_ROUND_BOUND_APRIORI=1: round 4)" if os.environ.get("TRX_ROUND_BOUND_APRIORI") else
                               "approx mode (bf16 rounding + listing slack from the measured rounding-error norms: round 5)"),
                      "K": st["k_split"], "search_ms": (t1 - t0) * 1e3, "scan_ms": st["scan_ms"], "queries_per_s": nq / (t1 - t0),
                      "n_rescored": st["n_rescored"], "n_rescanned": st["n_rescanned"], "n_uncertified": st["n_uncertified"],
                      "checksum_I": int(I.sum().item()) % 1000003, "checksum_D": float(D.double().sum().item())}), flush=True)
    del idx
