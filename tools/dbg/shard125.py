import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import textreact_amd.faiss_compat as faiss
dev = torch.device("cuda", 0)
queries = bench.make_rows(65536, 768, 5678, dev)
shard = bench.make_rows(125000, 768, 1234, dev)
idx = faiss.IndexFlatIP(768, device=0); idx.add(shard)
for timing in (False, True):
    idx.set_timing(timing)
    for _ in range(3): idx.search(queries, 10)
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); idx.search(queries, 10); b.record(); b.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print(os.environ.get("TRX_NO_BOOT"), os.environ.get("TRX_NO_RESCAN"), "timing", timing, "step median %.3f min %.3f" % (ts[6], ts[0]), "scan %.3f" % idx.last_stats()["scan_ms"], flush=True)
