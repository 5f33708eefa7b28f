"""attention forward through the launch wrapper with and without dropout at the predictor's shapes: python3 tools/attn_fwd_drop.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
import bench_predictor as bp
dev = "cuda"
for (B, H, Lq, Lk, causal, name) in ((32, 12, 512, 512, False, "self 512x512"), (32, 12, 160, 512, False, "cross 160x512"), (32, 12, 160, 160, True, "causal 160x160")):
    q, k, v = (torch.randn(B, L, H, 64, device=dev).to(torch.bfloat16) for L in (Lq, Lk, Lk))
    m = torch.zeros(B, Lk, device=dev)
    for p in (0.0, 0.1):
        ts = sorted(bp.timeit(lambda: ops._attention_fwd_launch(q, k, v, m, causal, 0.125, p, 7, True), iters=30) for _ in range(5))
        print(name, "p", p, "median %.1f us min %.1f us" % (ts[2] * 1e3, ts[0] * 1e3))
