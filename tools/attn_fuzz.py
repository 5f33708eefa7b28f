"""randomised shapes through the matrix-core attention (forward, both backward passes, dropout on / off) against the fp32
PyTorch statement under the same dropout decisions: python3 tools/attn_fuzz.py [n] [seed] [bf16|fp32]
(fp32: the v_mfma_f32_16x16x4_f32 kernels of round 5 -- attn_fwd_f32.h, attn_bwd_f32.h -- held to 2e-5 relative)"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
from oracle import nn_ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
f32 = len(sys.argv) > 3 and sys.argv[3] == "fp32"
cast = (lambda t: t) if f32 else (lambda t: t.bfloat16())
neg = torch.finfo(torch.float32).min
worst = 0.0
nfail = 0
for it in range(n):
    B, H = rng.choice([1, 2, 3]), rng.choice([1, 2, 5, 12])
    Lq = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 160, 255, 256, 257, 300, 512, 513])
    causal = rng.random() < 0.4
    Lk = Lq if (causal and rng.random() < 0.5) else rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 191, 192, 193, 256, 500, 512, 1023, 1024, 1025, 1100])
    if causal and Lk < Lq:
        Lk = Lq
    mode = rng.choice(["none", "key", "full"])
    p = rng.choice([0.0, 0.0, 0.1, 0.3])
    g = torch.Generator(device="cuda").manual_seed(it)
    q = cast(torch.randn(B, Lq, H, 64, device="cuda", generator=g))
    k = cast(torch.randn(B, Lk, H, 64, device="cuda", generator=g))
    v = cast(torch.randn(B, Lk, H, 64, device="cuda", generator=g))
    do = cast(torch.randn(B, Lq, H * 64, device="cuda", generator=g))
    m = None
    if mode == "key":
        keep = (torch.rand(B, Lk, device="cuda", generator=g) > 0.3).float(); keep[:, 0] = 1
        m = (1 - keep) * neg
    elif mode == "full":
        keep = (torch.rand(B, Lq, Lk, device="cuda", generator=g) > 0.3).float(); keep[:, :, 0] = 1
        m = (1 - keep) * neg
    qs, ks, vs = (t.clone().requires_grad_(True) for t in (q, k, v))
    o = ops.attention(qs, ks, vs, mask=m, causal=causal, dropout_p=p, seed=1000 + it)
    o.backward(do)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    r = nn_ref.attention(qr, kr, vr, mask=m, causal=causal, dropout_p=p, seed=1000 + it)
    r.backward(do.float())
    errs = []
    for name, a, c in (("o", o, r), ("dq", qs.grad, qr.grad), ("dk", ks.grad, kr.grad), ("dv", vs.grad, vr.grad)):
        e = float((a.detach().float() - c.detach()).abs().max()) / max(1.0, float(c.detach().abs().max()))
        errs.append((name, round(e, 4)))
        worst = max(worst, e) if Lk > 1 else worst
        # Lk == 1: P = 1, the exact gradients of q and k are 0 and what is left is the bf16 rounding of O inside
        # delta = dO . O (every flash-style backward has it); judged on an absolute scale there
        # (the noise of dk adds up over the Lq queries and grows with the 1 / (1 - p) scaling: ~ 0.01 sqrt(Lq) / (1 - p))
        tol = max(0.15, 0.012 * Lq ** 0.5 / (1 - p)) if (Lk == 1 and name in ("dq", "dk")) else 2.5e-2
        if f32:
            tol = 2e-5 * max(1.0, Lk ** 0.5 / 8)
        if not (e <= tol) or not bool(torch.isfinite(a.float()).all()):
            print("FAIL", it, dict(B=B, H=H, Lq=Lq, Lk=Lk, causal=causal, mode=mode, p=p), errs)
            nfail += 1
print(n, "cases,", nfail, "failures; worst relative error", round(worst, 4))
