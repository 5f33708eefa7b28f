import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import textreact_amd.faiss_compat as faiss
g = torch.Generator(device="cuda"); g.manual_seed(1)
for n, nq, k, metric in ((200000, 256, 10, "IP"), (200000, 1024, 10, "IP"), (200000, 1024, 20, "IP"), (200000, 1024, 10, "L2"), (3000, 1024, 10, "IP"), (200000, 1000, 10, "IP")):
    y = torch.randn((n, 768), generator=g, device="cuda").to(torch.bfloat16); x = torch.randn((nq, 768), generator=g, device="cuda").to(torch.bfloat16)
    idx = (faiss.IndexFlatIP if metric == "IP" else faiss.IndexFlatL2)(768); idx.add(y)
    D, I = idx.search(x, k)
    st = idx.last_stats()
    print(n, nq, k, metric, "splits", st["n_splits"], "rescored", st["n_rescored"], "rescanned", st["n_rescanned"])
