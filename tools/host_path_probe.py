"""The FAISS protocol with HOST arrays, as the reference's CLI calls it (retrieve_faiss.py:62-74): numpy fingerprints in, numpy
(D, I) out -- where the time goes between the dtype conversion on the host, PCIe and the GPU.
    python3 tools/host_path_probe.py [n [d [dtype]]]      dtype: int8 (Morgan bit vectors) | int64 (reaction difference counts) | float32
TRX_HOST_THREADS sets the library's worker threads (default min(cores, 32))."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import textreact_amd.faiss_compat as faiss
from textreact_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dt = sys.argv[3] if len(sys.argv) > 3 else "int8"
rng = np.random.default_rng(0)
x = (rng.random((n, d), dtype=np.float32) < 0.05).astype(dt)
if dt == "int64":
    x *= rng.integers(-3, 4, (n, d), dtype=np.int64)
def lap(f):
    t0 = time.perf_counter(); r = f(); return r, (time.perf_counter() - t0) * 1e3
idx = faiss.IndexFlatL2(d)
idx.add(x[:4096]); idx.search(x[:4096], 20)                      # warm: library, worker pool, pinned buffers, workspaces
_, t_conv = lap(lambda: np.ascontiguousarray(x, dtype=np.float32))
t_add = []
for _ in range(3):
    idx.reset()
    t_add.append(lap(lambda: idx.add(x))[1])
idx.search(x[:70000], 20)
t_search = []
for _ in range(3):
    (D, I), t = lap(lambda: idx.search(x, 20))
    t_search.append(t)
st = idx.last_stats()
import torch
xd = torch.from_numpy(x[:65536].astype(np.float32)).cuda()
idx.search(xd, 20); torch.cuda.synchronize()
_, t_dev = lap(lambda: (idx.search(xd, 20), torch.cuda.synchronize()))
print(json.dumps({"n": n, "d": d, "dtype": dt, "host_threads": _lib.lib().trx_host_threads(), "host_cores": len(os.sched_getaffinity(0)),
                  "numpy_convert_to_f32_ms": round(t_conv), "add_ms_host_array": [round(t, 1) for t in t_add],
                  "search_ms_host_arrays_all_rows": [round(t, 1) for t in t_search],
                  "queries_per_s_host_arrays": round(n / min(t_search) * 1e3), "input_GB_per_s_search": round(x.nbytes / min(t_search) / 1e6, 1),
                  "input_GB_per_s_add": round(x.nbytes / min(t_add) / 1e6, 1),
                  "search_ms_65536_device_resident": round(t_dev, 1),
                  "queries_per_s_device_resident": round(65536 / t_dev * 1e3), "int8_scan": st["int8_scan"], "self_found_first": bool((D[:, 0] == 0).all())}))
