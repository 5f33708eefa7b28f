"""A/B of attention-forward build variants in ONE process (interleaved rounds; the methodology rule for deltas of a few
percent): every libtrxnn*.so given on the command line is dlopen'ed and its trx_attention_fwd timed on the predictor's
three shapes, R rounds x (variants in turn), 20 launches each; median and minimum per variant.
    python3 tools/attn_ab.py name=path.so [name=path.so ...] > gpurun_out/attention_ab.json"""
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
    L.trx_attention_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp]
    libs[name] = L
dev = torch.device("cuda", 0)
shapes = [(32, 12, 512, 512, False, "encoder self-attention"), (32, 12, 160, 512, False, "cross-attention"),
          (32, 12, 160, 160, True, "decoder causal self-attention")]
rounds, iters = 7, 20
out = {"what": "trx_attention_fwd, bf16, key mask; %d interleaved rounds x %d launches per variant, HIP events" % (rounds, iters), "shapes": []}
for (B, H, Lq, Lk, causal, what) in shapes:
    g = torch.Generator(device=dev); g.manual_seed(0)
    q = torch.randn(B, Lq, H, 64, device=dev, generator=g).bfloat16()
    k = torch.randn(B, Lk, H, 64, device=dev, generator=g).bfloat16()
    v = torch.randn(B, Lk, H, 64, device=dev, generator=g).bfloat16()
    m = torch.zeros(B, Lk, device=dev)
    st = vp(torch.cuda.current_stream().cuda_stream)
    res, outs = {n: [] for n in libs}, {}
    for n, L in libs.items():
        o = torch.empty(B, Lq, H * 64, device=dev, dtype=torch.bfloat16)
        assert L.trx_attention_fwd(vp(q.data_ptr()), vp(k.data_ptr()), vp(v.data_ptr()), vp(m.data_ptr()), 1, int(causal), B, H, Lq, Lk, 0.125, 1,
                                   vp(o.data_ptr()), st) == 0
        outs[n] = o
    torch.cuda.synchronize()
    base = outs[next(iter(libs))].float()
    agree = {n: float((o.float() - base).abs().max()) for n, o in outs.items()}
    for r in range(rounds):
        for n, L in libs.items():
            o = outs[n]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                L.trx_attention_fwd(vp(q.data_ptr()), vp(k.data_ptr()), vp(v.data_ptr()), vp(m.data_ptr()), 1, int(causal), B, H, Lq, Lk, 0.125, 1,
                                    vp(o.data_ptr()), st)
            b.record(); b.synchronize()
            res[n].append(a.elapsed_time(b) / iters * 1e3)
    fl = 4.0 * B * H * Lq * Lk * 64 * (0.5 if causal else 1.0)
    row = {"what": what, "B": B, "H": H, "Lq": Lq, "Lk": Lk, "causal": causal, "variants": {}}
    for n, t in res.items():
        t = sorted(t)
        row["variants"][n] = {"us_median": t[len(t) // 2], "us_min": t[0], "frac_of_bf16_peak_at_median": fl / (t[len(t) // 2] * 1e-6) / 2.5e15,
                              "max_abs_diff_vs_first": agree[n]}
    out["shapes"].append(row)
print(json.dumps(out, indent=1))
