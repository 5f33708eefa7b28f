#!/usr/bin/env python3
"""Secondary benchmark (not the headline; bench.py is): the predictor kernels at the shapes of
scripts/train_RCR.sh (per-GPU batch 32, L = 512, hidden 768, 12 heads) against their rooflines.

  add+LayerNorm : HBM-bound, algorithmic bytes = rows*cols*(x + res + y)*sizeof(dtype)
  attention     : FLOPs = 4*B*H*Lq*Lk*64 (bf16: MFMA flash kernel, peak 2.5 PFLOP/s dense;
                  fp32: VALU kernel, the accuracy path, peak 157.3 TFLOP/s vector fp32)
Prints one JSON line per kernel; timing = torch.cuda.Event on torch's current stream, which is the
stream the kernels are launched on (ops.py passes it through the C ABI)."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from textreact_amd.predictor import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = "cuda"
    out = []
    for dtype, esz in ((torch.bfloat16, 2), (torch.float32, 4)):
        rows, cols = 32 * 512, 768
        x = torch.randn(rows, cols, device=dev).to(dtype); r = torch.randn(rows, cols, device=dev).to(dtype)
        g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev)
        ms = timeit(lambda: ops.add_layernorm(x, r, g, b, 1e-12))
        gbs = rows * cols * 3 * esz / (ms * 1e-3) / 1e9
        out.append({"kernel": "add_ln_fwd", "dtype": str(dtype), "rows": rows, "cols": cols, "ms": ms,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}})
        ref = timeit(lambda: torch.nn.functional.layer_norm(x + r, (cols,), g.to(dtype), b.to(dtype), 1e-12))
        out[-1]["torch_unfused_ms"] = ref
    for dtype in (torch.bfloat16, torch.float32):
        for (B, H, Lq, Lk, causal, name) in ((32, 12, 512, 512, False, "encoder self-attention"),
                                             (32, 12, 160, 512, False, "cross-attention"),
                                             (32, 12, 160, 160, True, "decoder causal self-attention")):
            q = torch.randn(B, Lq, H, 64, device=dev).to(dtype); k = torch.randn(B, Lk, H, 64, device=dev).to(dtype)
            v = torch.randn(B, Lk, H, 64, device=dev).to(dtype)
            m = torch.zeros(B, Lk, device=dev)
            ms = timeit(lambda: ops.attention(q, k, v, mask=m, causal=causal), iters=10)
            fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
            tf = fl / (ms * 1e-3) / 1e12
            ref = timeit(lambda: ops.attention(q, k, v, mask=m, causal=causal, backend="torch"), iters=10)
            out.append({"kernel": "attention_fwd", "what": name, "dtype": str(dtype), "B": B, "H": H, "Lq": Lq, "Lk": Lk,
                        "ms": ms, "torch_eager_fp32_ms": ref,
                        "roofline": ({"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0}
                                     if dtype == torch.bfloat16 else
                                     {"bound": "valu-fp32", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3})})
    # backward (bf16, matrix cores): dq pass + dk/dv pass + the per-query scalar prep; FLOPs counted as the
    # 5 GEMMs of the textbook backward (the two passes recompute S and dP, 7 GEMMs are executed)
    for (B, H, Lq, Lk, causal, name) in ((32, 12, 512, 512, False, "encoder self-attention"),
                                         (32, 12, 160, 512, False, "cross-attention"),
                                         (32, 12, 160, 160, True, "decoder causal self-attention")):
        dtype = torch.bfloat16
        q, k, v = (torch.randn(B, L, H, 64, device=dev).to(dtype).requires_grad_(True) for L in (Lq, Lk, Lk))
        m = torch.zeros(B, Lk, device=dev)
        o = ops.attention(q, k, v, mask=m, causal=causal)
        do = torch.randn_like(o)
        ms = timeit(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True), iters=10)
        fl = 10.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
        tf = fl / (ms * 1e-3) / 1e12
        qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
        orf = ops.attention(qr, kr, vr, mask=m, causal=causal, backend="torch")
        ref = timeit(lambda: torch.autograd.grad(orf, (qr, kr, vr), do.float(), retain_graph=True), iters=5)
        out.append({"kernel": "attention_bwd", "what": name, "dtype": str(dtype), "B": B, "H": H, "Lq": Lq, "Lk": Lk,
                    "ms": ms, "torch_eager_fp32_ms": ref,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0}})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
