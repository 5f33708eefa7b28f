"""round 5: the launches around the scan at C1's query count over a SMALL corpus (65,536 x 768 bf16 queries, 20,000 rows), five
searches -- under rocprofv3 --kernel-trace --stats this prices row_stats / select / classify without a 70 ms scan beside them"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import textreact_amd.faiss_compat as faiss
g = torch.Generator(device="cuda"); g.manual_seed(3)
y = torch.randn((20000, 768), generator=g, device="cuda").bfloat16()
x = torch.randn((65536, 768), generator=g, device="cuda").bfloat16()
idx = faiss.IndexFlatIP(768); idx.add(y)
for _ in range(5):
    D, I = idx.search(x, 10)
torch.cuda.synchronize()
print(int(I.sum()))
