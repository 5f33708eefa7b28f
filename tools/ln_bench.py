#!/usr/bin/env python3
"""add+LayerNorm kernels alone at the trainer's shapes (16384 x 768, 5120 x 768): forward and backward of the three storage
variants (bf16, fp32, mixed = bf16 x + fp32 residual stream; dropout 0.1 as in training) through the C ABI with operands
allocated once (an autograd call costs more host time than these kernels run), event-timed over 50 back-to-back launches.
Algorithmic bytes: forward x + res + y (+ the bf16 copy of y, mixed); backward dy + x + res in, dz + dx out (+ dy16, mixed).
python3 tools/ln_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor import ops
from textreact_amd.predictor.ops import lib, _p, _dt, _stream, _check


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def rows_for(rows):
    dev, cols, p, seed, eps = "cuda", 768, 0.1, 7, 1e-12
    L = lib()
    g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev)
    dg, db, dxb = (torch.empty(cols, device=dev) for _ in range(3))
    out = []
    if True:
        mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        nblk = L.trx_add_layernorm_bwd_blocks(rows)
        ws = torch.empty(3 * nblk * cols, device=dev)
        for name in ("bf16", "fp32", "mixed"):
            xd = torch.float32 if name == "fp32" else torch.bfloat16
            rd = torch.bfloat16 if name == "bf16" else torch.float32
            x = torch.randn(rows, cols, device=dev).to(xd); r = torch.randn(rows, cols, device=dev).to(rd)
            y, dy, dz, dx = torch.empty_like(r), torch.randn(rows, cols, device=dev).to(rd), torch.empty_like(r), torch.empty_like(x)
            es = lambda t: 2 if t == torch.bfloat16 else 4
            st = _stream(x)
            if name == "mixed":
                y16, dy16 = torch.empty_like(x), torch.randn(rows, cols, device=dev).to(torch.bfloat16)
                fwd = lambda: _check(L.trx_add_layernorm_fwd_mixed(_p(x), _p(r), _p(g), _p(b), eps, rows, cols, p, seed, _p(y), _p(y16), _p(mean), _p(rstd), None, st))
                bwd = lambda: _check(L.trx_add_layernorm_bwd_mixed(_p(dy), _p(dy16), _p(x), _p(r), _p(g), _p(mean), _p(rstd), rows, cols, p, seed,
                                                                    _p(dz), _p(dx), _p(dg), _p(db), None, None, _p(ws), st))
                fb, bb = rows * cols * (2 + 4 + 4 + 2), rows * cols * (4 + 2 + 2 + 4 + 4 + 2)
            else:
                fwd = lambda: _check(L.trx_add_layernorm_fwd_dropout(_p(x), _p(r), _p(g), _p(b), eps, rows, cols, _dt(x), p, seed, _p(y), _p(mean), _p(rstd), st))
                bwd = lambda: _check(L.trx_add_layernorm_bwd_dropout(_p(dy), _p(x), _p(r), _p(g), _p(mean), _p(rstd), rows, cols, _dt(x), p, seed,
                                                                      _p(dz), _p(dx), _p(dg), _p(db), _p(ws), st))
                fb, bb = rows * cols * 3 * es(xd), rows * cols * 5 * es(xd)
            fwd(); f = timeit(fwd); bw = timeit(bwd)
            out.append({"rows": rows, "variant": name, "fwd_us": round(f, 2), "fwd_GBs": round(fb / f / 1e3, 1), "fwd_frac": round(fb / f / 8e6, 3), "fwd_bytes": fb,
                        "bwd_us": round(bw, 2), "bwd_GBs": round(bb / bw / 1e3, 1), "bwd_frac": round(bb / bw / 8e6, 3), "bwd_bytes": bb})
            if name == "mixed":
                # what a call costs inside ops.backward() since round 5: the first stage only (dgamma = dbeta = NULL: the partial
                # column sums stay in ws; ONE trx_add_layernorm_bwd_reduce_many launch per backward pass finishes all calls)
                bwd1 = lambda: _check(L.trx_add_layernorm_bwd_mixed(_p(dy), _p(dy16), _p(x), _p(r), _p(g), _p(mean), _p(rstd), rows, cols, p, seed,
                                                                     _p(dz), _p(dx), None, None, None, None, _p(ws), st))
                b1 = timeit(bwd1)
                out.append({"rows": rows, "variant": "mixed, first stage only (the trainer's per-call cost: column sums deferred to one launch per pass)",
                            "fwd_us": round(f, 2), "fwd_GBs": round(fb / f / 1e3, 1), "fwd_frac": round(fb / f / 8e6, 3), "fwd_bytes": fb,
                            "bwd_us": round(b1, 2), "bwd_GBs": round(bb / b1 / 1e3, 1), "bwd_frac": round(bb / b1 / 8e6, 3), "bwd_bytes": bb})
    return out


def main():
    for rows in (16384, 5120):
        for o in rows_for(rows):
            print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
