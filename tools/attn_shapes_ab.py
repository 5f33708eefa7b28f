"""A/B of libtrxnn builds in ONE process (interleaved rounds): trx_attention_fwd_lse and trx_attention_bwd at the predictor's
five attention shapes -- encoder 512 x 512, RetroSyn's decoder (T = 160: cross 160 x 512, causal 160 x 160) and RCR's
(T = 7: cross 7 x 512, causal 7 x 7) -- B 32, 12 heads of 64, bf16, key mask.  Per variant: median / minimum of R rounds
x 20 launches (HIP events), the fraction of the dense bf16 MFMA peak (4 B H Lq Lk 64 forward, 10 ... backward, halved when
causal) and, for the byte-bound T = 7 shapes, of 8 TB/s on q + k + v + out (+ dout, dq, dk, dv backward); outputs compared
with the first variant's.
    python3 tools/attn_shapes_ab.py name=path.so [name=path.so ...] > gpurun_out/r04/attention_ab.json"""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

vp, i32, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
libs = {}
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    L = ctypes.CDLL(os.path.abspath(path))
    L.trx_attention_fwd_lse.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp]
    L.trx_attention_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp]
    libs[name] = L
dev = torch.device("cuda", 0)
shapes = [(32, 12, 512, 512, False, "encoder self-attention"), (32, 12, 160, 512, False, "cross-attention, T = 160"),
          (32, 12, 160, 160, True, "decoder causal self-attention, T = 160"), (32, 12, 7, 512, False, "cross-attention, T = 7"),
          (32, 12, 7, 7, True, "decoder causal self-attention, T = 7")]
rounds, iters = 7, 20
out = {"what": "trx_attention_fwd_lse / trx_attention_bwd, bf16, key mask; %d interleaved rounds x %d launches per variant, HIP events" % (rounds, iters),
       "shapes": []}
P = lambda t: vp(t.data_ptr())
for (B, H, Lq, Lk, causal, what) in shapes:
    g = torch.Generator(device=dev); g.manual_seed(0)
    q = torch.randn(B, Lq, H, 64, device=dev, generator=g).bfloat16()
    k = torch.randn(B, Lk, H, 64, device=dev, generator=g).bfloat16()
    v = torch.randn(B, Lk, H, 64, device=dev, generator=g).bfloat16()
    do = torch.randn(B, Lq, H * 64, device=dev, generator=g).bfloat16()
    m = torch.zeros(B, Lk, device=dev); m[::3, Lk * 4 // 5:] = -1e4
    st = vp(torch.cuda.current_stream().cuda_stream)
    bufs = {}
    for n, L in libs.items():
        o = torch.empty(B, Lq, H * 64, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, Lq, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        args_f = (P(q), P(k), P(v), P(m), 1, int(causal), B, H, Lq, Lk, 0.125, 1, P(o), P(lse), st)
        args_b = (P(q), P(k), P(v), P(m), 1, int(causal), B, H, Lq, Lk, 0.125, 1, P(o), P(do), P(lse), P(dq), P(dk), P(dv), st)
        assert L.trx_attention_fwd_lse(*args_f) == 0 and L.trx_attention_bwd(*args_b) == 0
        bufs[n] = (o, lse, dq, dk, dv, args_f, args_b)
    torch.cuda.synchronize()
    first = bufs[next(iter(libs))]
    agree = {n: [float((a.float() - b.float()).abs().max()) for a, b in zip(bb[:5], first[:5])] for n, bb in bufs.items()}
    res = {n: {"fwd": [], "bwd": []} for n in libs}
    for r in range(rounds):
        for n, L in libs.items():
            for which, fn, args in (("fwd", L.trx_attention_fwd_lse, bufs[n][5]), ("bwd", L.trx_attention_bwd, bufs[n][6])):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(iters):
                    fn(*args)
                b.record(); b.synchronize()
                res[n][which].append(a.elapsed_time(b) / iters * 1e3)
    fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
    by_f = 2.0 * B * H * 64 * (2 * Lq + 2 * Lk)                    # q, out + k, v
    by_b = 2.0 * B * H * 64 * (4 * Lq + 4 * Lk)                    # q, out, dout, dq + k, v, dk, dv
    row = {"what": what, "B": B, "H": H, "Lq": Lq, "Lk": Lk, "causal": causal, "variants": {}}
    for n in libs:
        tf, tb = sorted(res[n]["fwd"]), sorted(res[n]["bwd"])
        mf, mb = tf[len(tf) // 2], tb[len(tb) // 2]
        row["variants"][n] = {"fwd_us_median": mf, "fwd_us_min": tf[0], "bwd_us_median": mb, "bwd_us_min": tb[0],
                              "fwd_frac_of_bf16_peak": fl / (mf * 1e-6) / 2.5e15, "bwd_frac_of_bf16_peak": 2.5 * fl / (mb * 1e-6) / 2.5e15,
                              "fwd_frac_of_hbm": by_f / (mf * 1e-6) / 8e12, "bwd_frac_of_hbm": by_b / (mb * 1e-6) / 8e12,
                              "max_abs_diff_vs_first [out, lse, dq, dk, dv]": agree[n]}
    out["shapes"].append(row)
print(json.dumps(out, indent=1))
