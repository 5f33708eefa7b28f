"""round 5: the fp32 attention forward on the matrix cores (attn_fwd_f32.h) against the VALU kernel it replaces (TRX_NN_ATTN_VALU=1,
read once per process: run this script under both) and against the PyTorch-eager statement, at the predictor's three shapes;
also the largest absolute difference from the fp32 PyTorch statement (the <= 1e-3 parity path).
    python3 tools/r05/attn_f32_ab.py; TRX_NN_ATTN_VALU=1 python3 tools/r05/attn_f32_ab.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench_predictor import timeit  # noqa: E402
from oracle import nn_ref  # noqa: E402
from textreact_amd.predictor import ops  # noqa: E402

dev = "cuda"
form = "valu" if os.environ.get("TRX_NN_ATTN_VALU") else "mfma_f32"
for (B, H, Lq, Lk, causal, name) in ((32, 12, 512, 512, False, "encoder self-attention"), (32, 12, 160, 512, False, "cross-attention"),
                                     (32, 12, 160, 160, True, "decoder causal self-attention"), (32, 12, 7, 512, False, "cross-attention, 7 queries"),
                                     (3, 12, 437, 133, False, "ragged")):
    g = torch.Generator(device=dev); g.manual_seed(1)
    q, k, v = (torch.randn(B, L, H, 64, device=dev, generator=g) for L in (Lq, Lk, Lk))
    m = torch.zeros(B, Lk, device=dev); m[:, Lk - Lk // 5:] = -10000.0
    out = ops._attention_fwd_launch(q, k, v, m, causal, 0.125, 0.0, 0, True)[0]
    ref = nn_ref.attention(q, k, v, mask=m, causal=causal)
    err = float((out.reshape(ref.shape) - ref).abs().max())
    ms = timeit(lambda: ops._attention_fwd_launch(q, k, v, m, causal, 0.125, 0.0, 0, True), iters=30)
    eager = timeit(lambda: nn_ref.attention(q, k, v, mask=m, causal=causal), iters=10)
    fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
    print(json.dumps({"kernel": "attention_fwd fp32", "form": form, "what": name, "B": B, "H": H, "Lq": Lq, "Lk": Lk, "ms": ms, "torch_eager_fp32_ms": eager,
                      "max_abs_err_vs_torch_fp32": err, "TFLOPs": fl / (ms * 1e-3) / 1e12, "frac_of_157_TFLOPs": fl / (ms * 1e-3) / 1e12 / 157.3}), flush=True)

# backward (attn_bwd_f32.h against the VALU kernels, same switch): dq pass + dk/dv pass; FLOPs counted as the textbook 5 GEMMs
for (B, H, Lq, Lk, causal, name) in ((32, 12, 512, 512, False, "encoder self-attention"), (32, 12, 160, 512, False, "cross-attention"),
                                     (32, 12, 160, 160, True, "decoder causal self-attention"), (3, 12, 437, 133, False, "ragged")):
    g = torch.Generator(device=dev); g.manual_seed(2)
    q, k, v = (torch.randn(B, L, H, 64, device=dev, generator=g).requires_grad_(True) for L in (Lq, Lk, Lk))
    m = torch.zeros(B, Lk, device=dev); m[:, Lk - Lk // 5:] = -10000.0
    ref = nn_ref.attention(q, k, v, mask=m, causal=causal)
    do = torch.randn(ref.shape, device=dev, generator=g)
    want = torch.autograd.grad(ref, (q, k, v), do)
    with torch.no_grad():
        o_, lse_, mm_, mode_ = ops._attention_fwd_launch(q, k, v, m, causal, 0.125, 0.0, 0, True)
        got = ops._attention_bwd_launch(q, k, v, mm_, mode_, causal, 0.125, 0.0, 0, o_, do, lse_)
        err = max(float((a - b_).abs().max()) for a, b_ in zip(got, want))
        ms = timeit(lambda: ops._attention_bwd_launch(q, k, v, mm_, mode_, causal, 0.125, 0.0, 0, o_, do, lse_), iters=20)
    fl = 10.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
    print(json.dumps({"kernel": "attention_bwd fp32", "form": form, "what": name, "B": B, "H": H, "Lq": Lq, "Lk": Lk, "ms": ms,
                      "max_abs_err_vs_torch_autograd_fp32": err, "TFLOPs_on_the_5_textbook_GEMMs": fl / (ms * 1e-3) / 1e12}), flush=True)
