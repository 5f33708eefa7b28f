"""round 5: the search stage of an on-the-fly refresh on embeddings shaped like a randomly initialised BERT's [CLS] vectors
(40,000 queries x 204,800 passages x 768, bf16, IP, k = 10; every vector = one common direction + 15 % of its own: mean cosine
of pairs ~0.98), three searches: what the fall-back tiers (wide re-score, re-scan) cost on clustered data, kernel by kernel
under rocprofv3 --kernel-trace --stats"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import textreact_amd.faiss_compat as faiss
g = torch.Generator(device="cuda"); g.manual_seed(5)
c = torch.randn(768, generator=g, device="cuda")
def make(n):
    v = c[None, :] + 0.15 * 768 ** 0.5 / 768 ** 0.5 * torch.randn((n, 768), generator=g, device="cuda")
    return (v / v.norm(dim=1, keepdim=True) * 20).bfloat16()
y, x = make(204800), make(40000)
yy = y[:2000].float(); cos = (torch.nn.functional.normalize(yy, dim=1) @ torch.nn.functional.normalize(yy, dim=1).T).mean().item()
idx = faiss.IndexFlatIP(768); idx.add(y)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    D, I = idx.search(x, 10); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
st = idx.last_stats()
print("mean cosine %.3f; search %.2f ms; stats %s" % (cos, ms, {k: v for k, v in st.items() if k.startswith("n_") or k in ("int8_scan", "exact_class")}))
