"""attention backward (dq pass + dk/dv pass + prep) through the launch wrapper at the predictor's three shapes; run once per
library build (TRX_NN_LIB=...) on the same box: python3 tools/attn_bwd_ab.py [dropout p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
import bench_predictor as bp
dev = "cuda"
for (B, H, Lq, Lk, causal, name) in ((32, 12, 512, 512, False, "self 512x512"), (32, 12, 160, 512, False, "cross 160x512"), (32, 12, 160, 160, True, "causal 160x160")):
    q, k, v = (torch.randn(B, L, H, 64, device=dev).to(torch.bfloat16) for L in (Lq, Lk, Lk))
    m = torch.zeros(B, Lk, device=dev)
    P = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0      # dropout probability (training: 0.1)
    o, lse, mm, mode = ops._attention_fwd_launch(q, k, v, m, causal, 0.125, P, 7, True)
    do = torch.randn_like(o)
    ts = sorted(bp.timeit(lambda: ops._attention_bwd_launch(q, k, v, mm, mode, causal, 0.125, P, 7, o, do, lse), iters=30) for _ in range(5))
    dq, dk, dv = ops._attention_bwd_launch(q, k, v, mm, mode, causal, 0.125, P, 7, o, do, lse)
    print(name, "median %.1f us min %.1f us" % (ts[2] * 1e3, ts[0] * 1e3), "checksum %.6f" % float(dq.float().sum() + dk.float().sum() + dv.float().sum()))
