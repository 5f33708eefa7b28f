import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from textreact_amd.predictor import ops
from oracle import nn_ref
bf = torch.bfloat16
NEG = torch.finfo(torch.float32).min
def run(Lq, Lk, masked, val=NEG, tag=""):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    q = torch.randn(1, Lq, 1, 64, device="cuda", generator=g).to(bf); k = torch.randn(1, Lk, 1, 64, device="cuda", generator=g).to(bf)
    v = torch.randn(1, Lk, 1, 64, device="cuda", generator=g).to(bf)
    m = torch.zeros(1, Lk, device="cuda")
    for j in masked: m[0, j] = val
    o, lse, mm, mode = ops._attention_fwd_launch(q, k, v, m, False, 0.125, 0.0, 0, True)
    ref = nn_ref.attention(q.float(), k.float(), v.float(), mask=m, causal=False)
    print(tag, (Lq, Lk), "masked", (masked[0], masked[-1]) if len(masked) else None, "val %.3g" % val, "out nan", int(o.isnan().sum()), "lse nan", int(lse.isnan().sum()),
          "err %.4f" % float((o.float().nan_to_num(9.0) - ref).abs().max()), "lse[0:3]", lse.flatten()[:3].tolist())
run(20, 77, list(range(39, 77)), tag="A")
run(20, 77, list(range(64, 77)), tag="B tile1 all masked")
run(20, 77, [76], tag="C one")
run(20, 77, list(range(64, 77)), val=-1e4, tag="D tile1 masked -1e4")
run(20, 77, list(range(64, 77)), val=-1e20, tag="E -1e20")
run(40, 77, list(range(64, 77)), tag="F KS2")
run(128, 77, list(range(64, 77)), tag="G KS1")
run(20, 128, list(range(64, 128)), tag="H no tail")
run(20, 100, list(range(64, 100)), tag="I")
run(20, 300, list(range(64, 128)), tag="J tile1 masked, more tiles")
