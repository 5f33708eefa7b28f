"""The predictor's two hot kernels at the shapes the reference's scripts run them at, N forward + backward calls per
shape, in a fixed order: the target of profiles/run_profile_predictor.sh (rocprofv3 attributes the i-th block of N
dispatches of a kernel to the i-th shape of that kernel printed here).
  attention (B 32, H 12, bf16; scripts/train_RetroSyn_tf.sh:33, train_RCR.sh:30):
     encoder self 512 x 512 | cross 160 x 512 | decoder causal 160 x 160 | RCR decoder causal 7 x 7 | RCR cross 7 x 512
  add + LayerNorm (768 columns): 16384 rows (encoder) and 5120 rows (decoder), bf16 and fp32, forward + backward.
  weight-gradient GEMM: all 90 problems of a B32 . L512 . T160 training step as one grouped launch (what ops.backward runs),
     and the encoder's FFN-up shape as one split-contraction call (main launch + reduction)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, H = 32, 12
ATT = [("self 512x512", 512, 512, False), ("cross 160x512", 160, 512, False), ("causal 160x160", 160, 160, True),
       ("causal 7x7", 7, 7, True), ("cross 7x512", 7, 512, False)]
LN = [("16384 rows bf16", 16384, torch.bfloat16), ("5120 rows bf16", 5120, torch.bfloat16),
      ("16384 rows fp32", 16384, torch.float32), ("5120 rows fp32", 5120, torch.float32)]
g = torch.Generator(device="cuda").manual_seed(0)
plan = {"N": N, "attention": [], "add_ln": [], "gemm_tn_grouped": [], "gemm_tn_split": []}
for name, lq, lk, causal in ATT:
    q = torch.randn(B, lq, H, 64, device="cuda", generator=g).bfloat16().requires_grad_()
    k = torch.randn(B, lk, H, 64, device="cuda", generator=g).bfloat16().requires_grad_()
    v = torch.randn(B, lk, H, 64, device="cuda", generator=g).bfloat16().requires_grad_()
    mask = torch.zeros(B, lk, device="cuda")
    for _ in range(N):
        o = ops.attention(q, k, v, mask=mask, causal=causal)
        o.backward(torch.ones_like(o))
    torch.cuda.synchronize()
    plan["attention"].append({"shape": name, "Lq": lq, "Lk": lk, "causal": causal, "flops_fwd": 4.0 * B * H * lq * lk * 64 * (0.5 if causal else 1.0),
                              # algorithmic bytes (bf16): forward q, out + k, v; dq pass q, dout, dq + k, v; dk/dv pass q, dout + k, v, dk, dv
                              "bytes_fwd": 2.0 * B * H * 64 * (2 * lq + 2 * lk), "bytes_dq": 2.0 * B * H * 64 * (3 * lq + 2 * lk),
                              "bytes_dkv": 2.0 * B * H * 64 * (2 * lq + 4 * lk)})
for name, rows, dt in LN:
    x = torch.randn(rows, 768, device="cuda", generator=g).to(dt).requires_grad_()
    r = torch.randn(rows, 768, device="cuda", generator=g).to(dt).requires_grad_()
    gm = torch.ones(768, device="cuda", requires_grad=True); bt = torch.zeros(768, device="cuda", requires_grad=True)
    for _ in range(N):
        y = ops.add_layernorm(x, r, gm, bt, 1e-12)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    esz = 2 if dt == torch.bfloat16 else 4
    plan["add_ln"].append({"shape": name, "rows": rows, "bytes_fwd": rows * 768 * 3 * esz, "bytes_bwd": rows * 768 * 4 * esz})
enc = [(32 * 512, 2304, 768), (32 * 512, 768, 768), (32 * 512, 3072, 768), (32 * 512, 768, 3072)] * 12
dec = [(32 * 160, 2304, 768), (32 * 160, 768, 768), (32 * 160, 768, 768), (32 * 512, 1536, 768), (32 * 160, 768, 768), (32 * 160, 3072, 768),
       (32 * 160, 768, 3072)] * 6
ab = {sh: (torch.randn(sh[0], sh[1], device="cuda", generator=g).bfloat16(), torch.randn(sh[0], sh[2], device="cuda", generator=g).bfloat16()) for sh in set(enc + dec)}
probs = [(ab[sh][0], ab[sh][1], torch.empty(sh[1], sh[2], device="cuda"), torch.empty(sh[1], device="cuda")) for sh in enc + dec]
for _ in range(N):
    ops.gemm_tn_grouped(probs)
torch.cuda.synchronize()
fl = sum(2.0 * m * n * k for (m, n, k) in enc + dec)
plan["gemm_tn_grouped"].append({"shape": "90 problems of a training step", "flops": fl,
                                "bytes": sum(2.0 * m * (n + k) + 4.0 * n * k for (m, n, k) in enc + dec)})      # dY, X read once, dW written once
sh = (32 * 512, 3072, 768)
for _ in range(N):
    ops.gemm_tn(ab[sh][0], ab[sh][1], colsum=True, out_dtype=torch.float32)
torch.cuda.synchronize()
plan["gemm_tn_split"].append({"shape": "FFN up 16384 x 3072 x 768", "flops": 2.0 * sh[0] * sh[1] * sh[2], "bytes": 2.0 * sh[0] * (sh[1] + sh[2]) + 4.0 * sh[1] * sh[2]})
print("PLAN " + json.dumps(plan))
