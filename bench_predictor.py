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
from oracle import nn_ref          # the PyTorch-eager statement of the ops: the reference timed beside the kernels


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def train_step_bench(dev, steps=10, dtype=torch.bfloat16, L=512, T=160):
    """one optimisation step of the full-size predictor at the scripts' per-GPU shapes
    (train_RetroSyn_tf.sh: batch 128 over 4 GPUs -> 32, encoder L = 512, decoder T = 160; bf16 autocast
    like --precision 16-mixed, dropout 0.1, AdamW), forward + backward + optimizer, random-init weights."""
    from textreact_amd.predictor.model import Config
    from textreact_amd.predictor import train
    res = []
    B = 32
    g = torch.Generator().manual_seed(0)
    batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev),
             "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
             "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev),
             "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
    batch["attention_mask"][::3, L * 4 // 5:] = 0
    for backend in ("hip", "torch"):      # the product, then the reference statement (oracle/nn_ref.py)
        print("train_step_bench T=%d: %s" % (T, backend), file=sys.stderr, flush=True)
        enc = Config(vocab_size=31090)
        dec = Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True)
        torch.manual_seed(0)
        p = train.Predictor(enc, dec, mlm=False).to(dev).train()
        opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)

        scaler = torch.amp.GradScaler("cuda") if dtype == torch.float16 else None      # what Lightning's 16-mixed adds

        def step():
            with torch.autocast("cuda", dtype=dtype):
                loss, _ = p.training_step(batch)
            if scaler is None:
                loss.backward()
                opt.step()
            else:
                scaler.scale(loss).backward()
                scaler.step(opt); scaler.update()
            opt.zero_grad(set_to_none=True)
            return loss
        with nn_ref.implementation(backend):      # "torch": the PyTorch-eager statement of the ops timed beside the kernels
            ms = timeit(step, iters=steps, warm=4)   # the caching allocator is still growing during the first steps

        def fwd():
            with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
                return p.model(**batch)[0]
        p.eval()
        with nn_ref.implementation(backend):
            ms_eval = timeit(fwd, iters=steps, warm=2)
        p.train()
        tokens = B * (L + T)
        name = "bf16 autocast" if dtype == torch.bfloat16 else "fp16 autocast + GradScaler"
        res.append({"kernel": "train_step", "ops": backend, "dtype": name, "B": B, "L": L, "T": T,
                    "ms": ms, "tokens_per_s": tokens / (ms * 1e-3)})
        res.append({"kernel": "forward_eval", "ops": backend, "dtype": name, "B": B, "L": L, "T": T,
                    "ms": ms_eval, "tokens_per_s": tokens / (ms_eval * 1e-3)})
        del p, opt
        torch.cuda.empty_cache()
    # the same step captured once in a HIP graph and replayed (train.GraphedStep: device-side dropout seeds, capturable AdamW),
    # in a child process: the runtime switch it needs (train.GRAPH_RUNTIME_ENV) is read when the HIP runtime starts
    if dtype == torch.bfloat16:
        import subprocess
        env = dict(os.environ); env[train.GRAPH_RUNTIME_ENV[0]] = train.GRAPH_RUNTIME_ENV[1]
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-row", str(T), str(steps)], env=env,
                           capture_output=True, text=True, timeout=900)
        rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not rows:
            raise RuntimeError("graph row failed: " + r.stdout[-1000:] + r.stderr[-2000:])
        res.extend(rows)
    return res


def graph_row(dev, T, steps, L=512, B=32):
    """train_step_bench's step through train.GraphedStep: 3 eager steps, the capture, then replays"""
    from textreact_amd.predictor.model import Config
    from textreact_amd.predictor import train
    g = torch.Generator().manual_seed(0)
    batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev),
             "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
             "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev),
             "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
    batch["attention_mask"][::3, L * 4 // 5:] = 0
    enc = Config(vocab_size=31090)
    dec = Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True)
    torch.manual_seed(0)
    p = train.Predictor(enc, dec, mlm=False).to(dev).train()
    opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02, capturable=True)
    gs = train.GraphedStep(p, opt, max_grad_norm=None, autocast_dtype=torch.bfloat16)
    ms = timeit(lambda: gs.step(batch), iters=steps, warm=6)
    assert gs.replays >= steps, gs.replays
    gs.close()
    return {"kernel": "train_step", "ops": "hip, whole step replayed from one HIP graph", "dtype": "bf16 autocast", "B": B, "L": L, "T": T,
            "ms": ms, "tokens_per_s": B * (L + T) / (ms * 1e-3), "env": "%s=%s" % train.GRAPH_RUNTIME_ENV}


def generate_bench(dev, B=8):
    """test step (main.py:218-226): beam search with num_beams = 20 over max_dec_length = 160 (train_RetroSyn_tf.sh:33,43),
    full-size model, bf16 autocast, random-init weights (no end token wins early: every beam runs the full length)"""
    from textreact_amd.predictor.model import Config, TextReactModel
    from textreact_amd.predictor.generate import generate
    res = []
    L, nb, T = 512, 20, 160
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1, 31090, (B, L), generator=g).to(dev)
    am = torch.ones(B, L, dtype=torch.long, device=dev)
    for backend, graph in (("hip", True), ("hip", False), ("torch", False)):
        torch.manual_seed(0)
        m = TextReactModel(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1,
                                                            layer_norm_eps=1e-5, is_decoder=True)).to(dev).eval()

        def run():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return generate(m, ids, am, num_beams=nb, num_return_sequences=nb, max_length=T, length_penalty=0,
                                bos_token_id=12, eos_token_id=13, pad_token_id=0, graph=graph)
        with nn_ref.implementation(backend):
            ms = timeit(run, iters=2, warm=1)
        res.append({"kernel": "generate", "ops": backend, "decode_step": "hip graph replay" if graph else "eager launches",
                    "dtype": "bf16 autocast", "B": B, "L": L, "num_beams": nb,
                    "max_length": T, "ms": ms, "decoded_tokens_per_s": B * nb * (T - 1) / (ms * 1e-3)})
        del m
        torch.cuda.empty_cache()
    return res


def main():
    dev = "cuda"
    out = []
    # add+LayerNorm forward and backward through the C ABI (tools/ln_bench.py: operands allocated once), dropout 0.1 as in
    # training; "mixed" = bf16 x + fp32 residual stream, the variant the trainer runs under bf16 autocast
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import ln_bench
    for o in ln_bench.rows_for(16384):
        for which in ("fwd", "bwd"):
            out.append({"kernel": "add_ln_" + which, "variant": o["variant"], "rows": o["rows"], "cols": 768, "ms": o[which + "_us"] * 1e-3,
                        "includes": "the dgamma / dbeta reduction launch" if which == "bwd" else None,
                        "roofline": {"bound": "hbm", "achieved": o[which + "_GBs"], "peak": 8000.0, "unit": "GB/s", "frac": o[which + "_frac"],
                                     "algorithmic_bytes": o[which + "_bytes"]}})
    for dtype in (torch.bfloat16, torch.float32):
        for (B, H, Lq, Lk, causal, name) in ((32, 12, 512, 512, False, "encoder self-attention"),
                                             (32, 12, 160, 512, False, "cross-attention"),
                                             (32, 12, 160, 160, True, "decoder causal self-attention")):
            q = torch.randn(B, Lq, H, 64, device=dev).to(dtype); k = torch.randn(B, Lk, H, 64, device=dev).to(dtype)
            v = torch.randn(B, Lk, H, 64, device=dev).to(dtype)
            m = torch.zeros(B, Lk, device=dev)
            # the launch wrapper itself (operands contiguous, outputs allocated by torch's caching allocator): an autograd.Function
            # call costs more host time than the small shapes run on the GPU
            ms = timeit(lambda: ops._attention_fwd_launch(q, k, v, m, causal, 0.125, 0.0, 0, True), iters=30)
            fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
            tf = fl / (ms * 1e-3) / 1e12
            ref = timeit(lambda: nn_ref.attention(q, k, v, mask=m, causal=causal), iters=10)
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
        with torch.no_grad():
            o_, lse_, mm_, mode_ = ops._attention_fwd_launch(q, k, v, m, causal, 0.125, 0.0, 0, True)
            ms = timeit(lambda: ops._attention_bwd_launch(q, k, v, mm_, mode_, causal, 0.125, 0.0, 0, o_, do, lse_), iters=30)
        fl = 10.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
        tf = fl / (ms * 1e-3) / 1e12
        qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
        orf = nn_ref.attention(qr, kr, vr, mask=m, causal=causal)
        ref = timeit(lambda: torch.autograd.grad(orf, (qr, kr, vr), do.float(), retain_graph=True), iters=5)
        out.append({"kernel": "attention_bwd", "what": name, "dtype": str(dtype), "B": B, "H": H, "Lq": Lq, "Lk": Lk,
                    "ms": ms, "torch_eager_fp32_ms": ref,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0}})
    # weight-gradient GEMM dW = dY^T X (trx_gemm_tn_bf16: contraction over the 16,384 token rows split over workgroups + a fixed-order
    # reduction launch), at the encoder's four Linear shapes, against the library (torch.matmul on the transposed view)
    for (N, K, name) in ((2304, 768, "query+key+value"), (768, 768, "attention output"), (3072, 768, "FFN up"), (768, 3072, "FFN down")):
        M = 32 * 512
        dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        ms = timeit(lambda: ops.gemm_tn(dy, x, colsum=True, out_dtype=torch.float32), iters=20)
        ref = timeit(lambda: torch.matmul(dy.t(), x), iters=20)
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        out.append({"kernel": "gemm_tn (dW + db)", "what": name, "M": M, "N": N, "K": K, "ms": ms, "library_ms": ref,
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0}})
    for o in out:
        print(json.dumps(o), flush=True)
    for rows in (lambda: train_step_bench(dev),              # RetroSyn: decoder length 160 (train_RetroSyn_tf.sh:33)
                 lambda: train_step_bench(dev, T=7),         # RCR: BOS + 5 condition tokens + EOS (train_RCR.sh)
                 lambda: generate_bench(dev)):
        for o in rows():
            print(json.dumps(o), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--graph-row":
        print(json.dumps(graph_row(torch.device("cuda", 0), int(sys.argv[2]), int(sys.argv[3]))), flush=True)
    else:
        main()
