import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreact_amd.predictor import ops
from oracle import nn_ref
torch.manual_seed(0)
bf = torch.bfloat16
for (B, H, Lq, Lk, mask, causal) in [(2, 2, 20, 77, "key", False), (2, 2, 20, 77, "none", False), (2, 2, 20, 128, "key", False), (2, 2, 20, 64, "key", False),
                                     (2, 2, 7, 512, "key", False), (1, 1, 20, 77, "key", False)]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    q = torch.randn(B, Lq, H, 64, device="cuda", generator=g).to(bf); k = torch.randn(B, Lk, H, 64, device="cuda", generator=g).to(bf)
    v = torch.randn(B, Lk, H, 64, device="cuda", generator=g).to(bf); do = torch.randn(B, Lq, H * 64, device="cuda", generator=g).to(bf)
    m = None
    if mask == "key":
        keep = torch.ones(B, Lk, device="cuda"); keep[0, Lk // 2 + 1:] = 0
        m = (1 - keep) * torch.finfo(torch.float32).min
    o, lse, mm, mode = ops._attention_fwd_launch(q, k, v, m, causal, 0.125, 0.0, 0, True)
    dq, dk, dv = ops._attention_bwd_launch(q, k, v, mm, mode, causal, 0.125, 0.0, 0, o, do, lse)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = nn_ref.attention(qr, kr, vr, mask=m, causal=causal)
    gq, gk, gv = torch.autograd.grad(ref, (qr, kr, vr), do.float())
    print((B, H, Lq, Lk, mask, causal), "out nan", int(o.isnan().sum()), "lse nan", int(lse.isnan().sum()), "err out %.4f" % float((o.float() - ref).abs().max()),
          "| dq nan", int(dq.isnan().sum()), "dk nan", int(dk.isnan().sum()), "dv nan", int(dv.isnan().sum()),
          "| err dq %.4f dk %.4f dv %.4f" % tuple(float((a.float().nan_to_num(9.0) - b).abs().max()) for a, b in ((dq, gq), (dk, gk), (dv, gv))))
    if dq.isnan().any():
        idx = dq.isnan().nonzero()
        print("   first nan dq idx", idx[:3].tolist(), "rows with nan (b, i, h):", sorted(set((int(a), int(b_), int(c)) for a, b_, c, _ in idx.tolist()))[:12])
