"""Attention and residual-add + LayerNorm as torch ops backed by libtrxnn.so (include/trx_nn.h).

There is ONE implementation: raw device pointers into the C ABI on torch's current stream.  Every op raises
(TrxNNError) if the library or a GPU is missing -- no fallback.  The plain fp32 PyTorch statement of the same ops,
which the kernel tests compare against and which lets the module tree be checked on a box without a GPU, is test
infrastructure and lives in oracle/nn_ref.py; nothing in this package imports it.
"""
import ctypes
import math
import os
import struct

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "csrc", os.environ.get("TRX_NN_LIB", "libtrxnn.so"))   # TRX_NN_LIB: diagnostic builds
_lib = None
F32, BF16 = 0, 1
MASK_NONE, MASK_KEY, MASK_FULL = 0, 1, 2
SYMBOLS = ["trx_add_layernorm_fwd", "trx_add_layernorm_bwd", "trx_add_layernorm_bwd_blocks",
           "trx_attention_fwd", "trx_attention_fwd_lse", "trx_attention_bwd",
           "trx_add_layernorm_fwd_dropout", "trx_add_layernorm_bwd_dropout", "trx_attention_fwd_dropout",
           "trx_attention_bwd_dropout", "trx_dropout_keep_mask", "trx_add_layernorm_fwd_mixed",
           "trx_add_layernorm_bwd_mixed", "trx_add_layernorm_bwd_reduce_many", "trx_attention_fwd_kvcache", "trx_attention_fwd_strided",
           "trx_attention_bwd_strided", "trx_attention_decode_gather", "trx_gemm_tn_bf16", "trx_gemm_tn_ws_bytes", "trx_gemm_tn_grouped_block_bytes", "trx_gemm_tn_grouped_plan", "trx_gemm_tn_grouped_run", "trx_gemm_tn_grouped", "trx_nn_last_error", "trx_nn_version",
           "trx_nn_set_seed_device", "trx_attention_bwd_ws", "trx_attention_bwd_ws_bytes"]


class TrxNNError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise TrxNNError("libtrxnn.so is missing: run `make -C textreact_amd/csrc`")
        L = ctypes.CDLL(_SO)
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        L.trx_add_layernorm_fwd.argtypes = [vp, vp, vp, vp, f32, i64, i32, i32, vp, vp, vp, vp]
        L.trx_add_layernorm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp]
        L.trx_add_layernorm_bwd_blocks.argtypes = [i64]
        L.trx_attention_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp]
        L.trx_attention_fwd_lse.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp]
        L.trx_attention_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp]
        u64 = ctypes.c_uint64
        L.trx_add_layernorm_fwd_dropout.argtypes = [vp, vp, vp, vp, f32, i64, i32, i32, f32, u64, vp, vp, vp, vp]
        L.trx_add_layernorm_bwd_dropout.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, f32, u64, vp, vp, vp, vp, vp, vp]
        L.trx_attention_fwd_dropout.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, f32, u64, vp, vp, vp]
        L.trx_attention_bwd_dropout.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, i32, f32, u64,
                                                vp, vp, vp, vp, vp, vp, vp]
        L.trx_dropout_keep_mask.argtypes = [u64, f32, i64, i64, i64, vp, vp]
        L.trx_add_layernorm_fwd_mixed.argtypes = [vp, vp, vp, vp, f32, i64, i32, f32, u64, vp, vp, vp, vp, vp, vp]
        L.trx_add_layernorm_bwd_mixed.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, u64, vp, vp, vp, vp, vp, vp, vp, vp]
        L.trx_add_layernorm_bwd_reduce_many.argtypes = [vp, i32, i32, vp]
        L.trx_attention_fwd_kvcache.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, f32, i32, vp, vp]
        L.trx_attention_fwd_strided.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, u64, vp, vp, vp]
        L.trx_attention_bwd_strided.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, u64,
                                                vp, vp, vp, vp, vp, vp, vp]
        L.trx_attention_decode_gather.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp]
        L.trx_gemm_tn_ws_bytes.argtypes = [i32, i32, i32]
        L.trx_gemm_tn_ws_bytes.restype = i64
        L.trx_gemm_tn_bf16.argtypes = [vp, i32, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, vp]
        L.trx_gemm_tn_grouped_block_bytes.argtypes = [vp, i32]
        L.trx_gemm_tn_grouped_block_bytes.restype = i64
        L.trx_gemm_tn_grouped_plan.argtypes = [vp, i32, vp, i64]
        L.trx_gemm_tn_grouped_run.argtypes = [vp, vp, vp]
        L.trx_gemm_tn_grouped.argtypes = [vp, i32, vp, i64, vp]
        L.trx_nn_set_seed_device.argtypes = [vp]
        L.trx_attention_bwd_ws_bytes.argtypes = [i32, i32, i32]
        L.trx_attention_bwd_ws_bytes.restype = i64
        L.trx_attention_bwd_ws.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, i32, f32, u64,
                                           vp, vp, vp, vp, vp, vp, vp, vp]
        L.trx_nn_last_error.restype = ctypes.c_char_p
        L.trx_nn_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise TrxNNError("trxnn error %d: %s" % (rc, lib().trx_nn_last_error().decode()))


def _need_gpu(t):
    if not t.is_cuda:
        raise TrxNNError("the HIP ops need tensors on a GPU (there is no CPU implementation behind them)")


def require_device(device):
    """the trainer / encoder entry points call this first: there is no CPU implementation behind the ops"""
    if torch.device(device).type != "cuda":
        raise TrxNNError("textreact_amd.predictor needs a GPU: attention and add+LayerNorm exist as HIP kernels only "
                         "(libtrxnn.so), there is no CPU implementation behind them")
    lib()


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TrxNNError("unsupported dtype %s (float32 / bfloat16)" % t.dtype)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    # the training step is host-bound: the raw-stream query is ~10x cheaper than building a Stream
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_seed_state = {"base": None, "n": 0, "device": None, "free": 0}
_M64 = (1 << 64) - 1


def _draw_seed():
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def set_seed_device(t):
    """device-side seed source (include/trx_nn.h: trx_nn_set_seed_device): `t` = an int64 tensor [1] on the GPU, or None to
    go back to host seeds.  While set, a dropout site's `seed` is just its number inside the forward pass (no draw from the
    host generator, the same arguments every step) and the kernels mix it with the value in `t`, which the training step
    bumps on the stream: the step can then be captured in a HIP graph and replayed (train.GraphedStep)."""
    if t is not None and not (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
        raise TrxNNError("set_seed_device: an int64 tensor with one element on the GPU")
    _check(lib().trx_nn_set_seed_device(_p(t)))
    _seed_state["device"] = t


class seed_scope:
    """`with ops.seed_scope():` around a forward pass: ONE draw from torch's CPU generator, and every dropout site
    inside takes a seed derived from it by a hashed counter (splitmix64).  torch.manual_seed therefore still makes
    runs repeatable, at ~1 us per site instead of the ~10 us of a generator draw (a training step has ~70 sites and
    is host-bound).  Outside a scope every site draws from the generator itself."""

    def __enter__(self):
        self.prev = (_seed_state["base"], _seed_state["n"])
        # device seeds: the sites of a pass are numbered 1, 2, ... (no host randomness: the pass may be a graph capture)
        _seed_state["base"], _seed_state["n"] = (0 if _seed_state["device"] is not None else _draw_seed()), 0
        return self

    def __exit__(self, *exc):
        _seed_state["base"], _seed_state["n"] = self.prev
        return False


def new_seed():
    """a fresh 62-bit dropout seed (see seed_scope)"""
    st = _seed_state
    if st["device"] is not None:
        if st["base"] is None:          # outside a scope: still a distinct site number
            st["free"] += 1
            return (1 << 40) + st["free"]
        st["n"] += 1
        return st["n"]
    if st["base"] is None:
        return _draw_seed()
    st["n"] += 1
    z = (st["base"] + st["n"] * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return (z ^ (z >> 31)) >> 2


def dropout_keep_mask(seed, p, streams, rows, cols, device):
    """bool [streams, rows, cols]: the decisions the kernels take for (seed, p), materialised -- so that a reference
    statement of an op (oracle/nn_ref.py) can drop the same elements in the parity tests"""
    keep = torch.empty((streams, rows, cols), dtype=torch.uint8, device=device)
    if not keep.is_cuda:
        raise TrxNNError("dropout decisions are computed on the GPU (there is no CPU implementation)")
    _check(lib().trx_dropout_keep_mask(int(seed), float(p), streams, rows, cols, _p(keep), _stream(keep)))
    return keep.bool()


def _add_ln_fwd_launch(x, res, gamma, beta, eps, p, seed, want_stats):
    """trx_add_layernorm_fwd_dropout: y = LayerNorm(dropout(x) + res); returns y, mean, rstd (None unless want_stats) and
    the operands as the backward takes them"""
    _need_gpu(x)
    xs = x.contiguous()
    rs = res.contiguous() if res is not None else None
    cols = xs.shape[-1]
    rows = xs.numel() // cols
    y = torch.empty_like(xs)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    g, b = gamma.float().contiguous(), beta.float().contiguous()
    _check(lib().trx_add_layernorm_fwd_dropout(_p(xs), _p(rs), _p(g), _p(b), float(eps), rows, cols, _dt(xs),
                                               float(p), int(seed), _p(y), _p(mean), _p(rstd), _stream(xs)))
    return y, mean, rstd, xs, rs, g


def _add_ln_bwd_launch(dy, xs, rs, g, mean, rstd, p, seed):
    """returns dz (the gradient of dropout(x) + res, = the residual's gradient), dx (x's gradient through its dropout; dz
    itself when p == 0), dgamma, dbeta"""
    dy = dy.contiguous()
    cols = xs.shape[-1]
    rows = xs.numel() // cols
    nblk = lib().trx_add_layernorm_bwd_blocks(rows)
    ws = torch.empty(2 * nblk * cols, dtype=torch.float32, device=xs.device)
    dz = torch.empty_like(xs)
    dx = torch.empty_like(xs) if p > 0 else None
    dg = torch.empty(cols, dtype=torch.float32, device=xs.device)
    db = torch.empty(cols, dtype=torch.float32, device=xs.device)
    _check(lib().trx_add_layernorm_bwd_dropout(_p(dy), _p(xs), _p(rs), _p(g), _p(mean), _p(rstd), rows, cols, _dt(xs),
                                               float(p), int(seed), _p(dz), _p(dx), _p(dg), _p(db), _p(ws), _stream(xs)))
    return dz, (dx if p > 0 else dz), dg, db


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p, seed):
        need = x.requires_grad or (res is not None and res.requires_grad) or gamma.requires_grad
        y, mean, rstd, xs, rs, g = _add_ln_fwd_launch(x, res, gamma, beta, eps, p, seed, need)
        if need:
            ctx.save_for_backward(xs, rs if rs is not None else xs.new_empty(0), g, mean, rstd)
            ctx.has_res = rs is not None
            ctx.drop = (float(p), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, rs, g, mean, rstd = ctx.saved_tensors
        rs = rs if ctx.has_res else None
        p, seed = ctx.drop
        dz, dx, dg, db = _add_ln_bwd_launch(dy, xs, rs, g, mean, rstd, p, seed)
        return dx, (dz if ctx.has_res else None), dg, db, None, None, None


class _AddLayerNormMixed(torch.autograd.Function):
    """x bf16 (a dense output under autocast), res / y fp32 (the residual stream); with `dual` also a bf16
    copy of y for the Linear layers that read it next (the cast autocast would run per use), whose gradient
    comes back as a second argument of backward and is summed inside the kernel"""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p, seed, dual, bias):
        _need_gpu(x)
        xs, rs = x.contiguous(), res.contiguous()
        xb = bias.float().contiguous() if bias is not None else None
        cols = xs.shape[-1]
        rows = xs.numel() // cols
        y = torch.empty_like(rs)
        y16 = torch.empty_like(xs) if dual else None
        need = x.requires_grad or res.requires_grad or gamma.requires_grad or (bias is not None and bias.requires_grad)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need else None
        g, b = gamma.float().contiguous(), beta.float().contiguous()
        _check(lib().trx_add_layernorm_fwd_mixed(_p(xs), _p(rs), _p(g), _p(b), float(eps), rows, cols, float(p), int(seed),
                                                 _p(y), _p(y16), _p(mean), _p(rstd), _p(xb), _stream(xs)))
        ctx.set_materialize_grads(False)
        if need:
            ctx.save_for_backward(xs, rs, g, mean, rstd, xb if xb is not None else xs.new_empty(0))
            ctx.drop = (float(p), int(seed), xb is not None)
            ctx.params = (gamma, beta, bias)       # what a deferred second stage hands its sums to (ops.backward)
        return (y, y16) if dual else y

    @staticmethod
    def backward(ctx, dy, dy16=None):
        xs, rs, g, mean, rstd, xb = ctx.saved_tensors
        p, seed, has_bias = ctx.drop
        xb = xb if has_bias else None
        cols = xs.shape[-1]
        rows = xs.numel() // cols
        if dy is None and dy16 is None:
            return None, None, None, None, None, None, None, None, None
        dy = dy.float().contiguous() if dy is not None else None
        dy16 = dy16.to(torch.bfloat16).contiguous() if dy16 is not None else None
        nblk = lib().trx_add_layernorm_bwd_blocks(rows)
        ws = torch.empty((3 if has_bias else 2) * nblk * cols, dtype=torch.float32, device=xs.device)
        dz, dx = torch.empty_like(rs), torch.empty_like(xs)
        if _deferred_ln is not None and _ln_deferrable(ctx.params, ctx.needs_input_grad):
            # inside ops.backward(): the first stage only; the column sums of every such call of the pass are finished by ONE
            # launch at the end (deferred_wgrad.__exit__), which puts them into the parameters' .grad directly
            _check(lib().trx_add_layernorm_bwd_mixed(_p(dy), _p(dy16), _p(xs), _p(rs), _p(g), _p(mean), _p(rstd), rows, cols, p, seed,
                                                     _p(dz), _p(dx), None, None, _p(xb), None, _p(ws), _stream(xs)))
            _deferred_ln.append((ws, nblk, cols, ctx.params))
            return dx, dz, None, None, None, None, None, None, None
        dg = torch.empty(cols, dtype=torch.float32, device=xs.device)
        db = torch.empty(cols, dtype=torch.float32, device=xs.device)
        dxb = torch.empty(cols, dtype=torch.float32, device=xs.device) if has_bias else None
        _check(lib().trx_add_layernorm_bwd_mixed(_p(dy), _p(dy16), _p(xs), _p(rs), _p(g), _p(mean), _p(rstd), rows, cols, p, seed,
                                                 _p(dz), _p(dx), _p(dg), _p(db), _p(xb), _p(dxb), _p(ws), _stream(xs)))
        return dx, dz, dg, db, None, None, None, None, dxb


def add_layernorm(x, res, gamma, beta, eps, dropout_p=0.0, seed=None, dual=False, bias=None):
    """LayerNorm(dropout(x) + res) * gamma + beta over the last dimension; res may be None.
    dropout_p > 0 (training): x is dropped before the residual is added, as BertSelfOutput /
    BertOutput / BertEmbeddings do; `seed` picks the decisions (default: a fresh one).
    dual=True returns (y, y_low): y_low is a bf16 copy of y written by the same kernel when x is bf16 and the
    residual stream fp32 (autocast) -- for the Linear layers that read y next -- and y itself otherwise.
    bias: the bias of the Linear that produced x, when the caller ran that Linear WITHOUT it: x + bias is formed
    here, and in the mixed-storage path inside the kernel, whose backward then yields the bias gradient too."""
    _need_gpu(x)
    if x.dtype == torch.float16:
        # fp16 autocast (the reference's --precision 16-mixed): the kernels store bf16 or fp32 -- run on bf16, hand fp16 back
        # where the caller would have got x's type
        out = add_layernorm(x.to(torch.bfloat16), res.to(torch.bfloat16) if (res is not None and res.dtype == torch.float16) else res,
                            gamma, beta, eps, dropout_p=dropout_p, seed=seed, dual=dual, bias=bias)
        if dual:
            return out[0].to(torch.float16) if out[0].dtype == torch.bfloat16 else out[0], out[1]
        return out.to(torch.float16) if out.dtype == torch.bfloat16 else out

    def _mixed(x, res):
        cols = x.shape[-1]
        return (res is not None and x.dtype == torch.bfloat16 and res.dtype == torch.float32
                and cols % 4 == 0 and cols <= 1024)
    if bias is not None and not _mixed(x, res):
        x, bias = x + bias.to(x.dtype), None
    if dual:
        if _mixed(x, res):
            if dropout_p > 0 and seed is None:
                seed = new_seed()
            return _AddLayerNormMixed.apply(x, res, gamma, beta, eps, float(dropout_p), 0 if seed is None else seed, True, bias)
        y = add_layernorm(x, res, gamma, beta, eps, dropout_p=dropout_p, seed=seed)
        return y, y
    if dropout_p > 0 and seed is None:
        seed = new_seed()
    if res is not None and res.dtype != x.dtype:
        # autocast: dense outputs are bf16 while the residual stream stays fp32 (torch runs layer_norm in
        # fp32 under autocast, so does the reference): the mixed kernels read x as bf16 and keep the
        # stream in fp32; anything else is widened to one storage type first.
        if _mixed(x, res):
            return _AddLayerNormMixed.apply(x, res, gamma, beta, eps, float(dropout_p), 0 if seed is None else seed, False, bias)
        wide = torch.promote_types(x.dtype, res.dtype)
        x, res = x.to(wide), res.to(wide)
    return _AddLayerNorm.apply(x, res, gamma, beta, eps, float(dropout_p), 0 if seed is None else seed)


def _attention_fwd_launch(q, k, v, mask, causal, scale, p, seed, want_lse):
    """trx_attention_fwd_dropout on contiguous [B, L, H, 64] operands; mask None | [B, Lk] | [B, Lq, Lk] (additive, fp32).
    Returns out [B, Lq, H*64], lse [B, H, Lq] (None unless want_lse), and the mask as the kernels take it."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    mode, m = MASK_NONE, None
    if mask is not None:
        m = mask.float().contiguous()
        mode = MASK_KEY if m.dim() == 2 else MASK_FULL
        assert m.shape == ((B, Lk) if mode == MASK_KEY else (B, Lq, Lk)), m.shape
    out = torch.empty((B, Lq, H * D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device) if want_lse else None
    _check(lib().trx_attention_fwd_dropout(_p(q), _p(k), _p(v), _p(m), mode, 1 if causal else 0, B, H, Lq, Lk,
                                           float(scale), _dt(q), float(p), int(seed), _p(out), _p(lse), _stream(q)))
    return out, lse, m, mode


def _attention_bwd_launch(q, k, v, m, mode, causal, scale, p, seed, out, dout, lse):
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    dout = dout.contiguous()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(2 * B * H * Lq, dtype=torch.float32, device=q.device)       # the caller's scratch: capturable
    _check(lib().trx_attention_bwd_ws(_p(q), _p(k), _p(v), _p(m) if mode != MASK_NONE else None, mode,
                                      1 if causal else 0, B, H, Lq, Lk, 0, 0, float(scale), _dt(q), float(p), int(seed),
                                      _p(out), _p(dout), _p(lse), _p(dq), _p(dk), _p(dv), _p(ws), _stream(q)))
    return dq, dk, dv


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, causal, scale, p, seed):
        need = q.requires_grad or k.requires_grad or v.requires_grad   # before .contiguous(): a copy made here has no flag
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, lse, m, mode = _attention_fwd_launch(q, k, v, mask, causal, scale, p, seed, need)
        if need:
            ctx.save_for_backward(q, k, v, m if m is not None else q.new_empty(0), out, lse)
            ctx.cfg = (mode, causal, scale, float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, m, out, lse = ctx.saved_tensors
        mode, causal, scale, p, seed = ctx.cfg
        dq, dk, dv = _attention_bwd_launch(q, k, v, m, mode, causal, scale, p, seed, out, dout, lse)
        return dq, dk, dv, None, None, None, None, None


def attention(q, k, v, mask=None, causal=False, scale=None, dropout_p=0.0, seed=None):
    """q [B, Lq, H, 64], k / v [B, Lk, H, 64] -> [B, Lq, H*64].
    mask: additive float, [B, Lk] (key padding) or [B, Lq, Lk]; causal: key j visible iff
    j <= i + (Lk - Lq).  dropout_p > 0 (training): dropout on the softmax probabilities
    (BertSelfAttention.dropout); `seed` picks the decisions (default: a fresh one)."""
    B, Lq, H, D = q.shape
    Lk = k.shape[1]
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if dropout_p > 0 and seed is None:
        seed = new_seed()
    _need_gpu(q)
    if q.dtype == torch.float16:   # fp16 autocast: on the bf16 kernels, fp16 handed back
        bf = torch.bfloat16
        return attention(q.to(bf), k.to(bf), v.to(bf), mask=mask, causal=causal, scale=scale,
                         dropout_p=dropout_p, seed=seed).to(torch.float16)
    if D != 64:
        raise TrxNNError("the attention kernel is specialised for heads of 64 (got %d)" % D)
    if (not (q.requires_grad or k.requires_grad or v.requires_grad) and dropout_p == 0 and not k.is_contiguous()
            and k.stride() == v.stride() and k.stride()[1:] == (H * D, D, 1) and k.stride(0) >= Lk * H * D):
        # the first Lk positions of a key/value cache [B, Lmax, H, 64]: read in place (decoding)
        qc = q.contiguous()
        mode, m = MASK_NONE, None
        if mask is not None:
            m = mask.float().contiguous()
            mode = MASK_KEY if m.dim() == 2 else MASK_FULL
        out = torch.empty((B, Lq, H * D), dtype=q.dtype, device=q.device)
        _check(lib().trx_attention_fwd_kvcache(_p(qc), _p(k), _p(v), _p(m), mode, 1 if causal else 0, B, H, Lq, Lk,
                                               k.stride(0), float(scale), _dt(qc), _p(out), _stream(qc)))
        return out
    return _Attention.apply(q, k, v, mask, causal, scale, float(dropout_p), 0 if seed is None else seed)


def attention_decode_gather(q, kv, anc, t_dev, scale=None):
    """one-token self-attention of beam search over a cache that is never re-ordered (include/trx_nn.h:
    trx_attention_decode_gather).  q [n, 1, H, 64] or [n, H, 64] bf16 (rows may be strided), kv [n, T, 2, H, 64] bf16,
    anc [n, T] int32 ancestor table, t_dev int64 [1] on the device = the last filled position.  -> [n, 1, H*64]"""
    _need_gpu(q)
    n, T, _, H, D = kv.shape
    if D != 64 or q.dtype != torch.bfloat16 or kv.dtype != torch.bfloat16 or anc.dtype != torch.int32 or t_dev.dtype != torch.int64:
        raise TrxNNError("attention_decode_gather: bf16 q / kv with heads of 64, int32 table, int64 position")
    q3 = q.reshape(n, H, D) if q.dim() == 4 else q
    if q3.stride(2) != 1 or q3.stride(1) != D:
        q3 = q3.contiguous()
    out = torch.empty((n, 1, H * D), dtype=torch.bfloat16, device=q.device)
    _check(lib().trx_attention_decode_gather(_p(q3), q3.stride(0), _p(kv.contiguous()), _p(anc.contiguous()), _p(t_dev), _p(out), n, H, T,
                                             float(scale if scale is not None else 1.0 / math.sqrt(D)), _stream(q)))
    return out


# ---- packed projections: q, k, v as slices of ONE GEMM output (bf16, matrix-core kernels) -------------------
def _mask_args(mask, B, Lq, Lk):
    if mask is None:
        return MASK_NONE, None
    m = mask.float().contiguous()
    mode = MASK_KEY if m.dim() == 2 else MASK_FULL
    assert m.shape == ((B, Lk) if mode == MASK_KEY else (B, Lq, Lk)), m.shape
    return mode, m


class _AttentionPacked(torch.autograd.Function):
    """`a` holds q (and k, v when `kv` is None): a = [B, L, 3, H, 64] for self-attention; otherwise a = q
    [B, Lq, H, 64] and kv = [B, Lk, 2, H, 64].  The kernels read the slices in place (row strides) and the
    backward writes dq, dk, dv straight into gradients of the same packed shapes."""

    @staticmethod
    def forward(ctx, a, kv, mask, causal, scale, p, seed):
        need = a.requires_grad or (kv is not None and kv.requires_grad)   # before .contiguous(): a copy has no flag
        a = a.contiguous()
        H, D = a.shape[-2], a.shape[-1]
        es = a.element_size()
        if kv is None:
            B, Lq = a.shape[0], a.shape[1]
            Lk, ldq, ldk = Lq, 3 * H * D, 3 * H * D
            qp, kp, vp = a.data_ptr(), a.data_ptr() + H * D * es, a.data_ptr() + 2 * H * D * es
        else:
            kv = kv.contiguous()
            B, Lq, Lk = a.shape[0], a.shape[1], kv.shape[1]
            ldq, ldk = H * D, 2 * H * D
            qp, kp, vp = a.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * D * es
        mode, m = _mask_args(mask, B, Lq, Lk)
        out = torch.empty((B, Lq, H * D), dtype=a.dtype, device=a.device)
        lse = torch.empty((B, H, Lq), dtype=torch.float32, device=a.device) if need else None
        c = ctypes.c_void_p
        _check(lib().trx_attention_fwd_strided(c(qp), c(kp), c(vp), _p(m), mode, 1 if causal else 0, B, H, Lq, Lk, ldq, ldk,
                                               float(scale), float(p), int(seed), _p(out), _p(lse), _stream(a)))
        if need:
            ctx.save_for_backward(a, kv if kv is not None else a.new_empty(0), m if m is not None else a.new_empty(0), out, lse)
            ctx.cfg = (kv is not None, mode, causal, scale, float(p), int(seed), B, H, D, Lq, Lk, ldq, ldk)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, kv, m, out, lse = ctx.saved_tensors
        has_kv, mode, causal, scale, p, seed, B, H, D, Lq, Lk, ldq, ldk = ctx.cfg
        es = a.element_size()
        dout = dout.contiguous()
        da = torch.empty_like(a)
        c = ctypes.c_void_p
        if has_kv:
            dkv = torch.empty_like(kv)
            ptrs = (a.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * D * es, da.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + H * D * es)
        else:
            dkv = None
            ptrs = (a.data_ptr(), a.data_ptr() + H * D * es, a.data_ptr() + 2 * H * D * es,
                    da.data_ptr(), da.data_ptr() + H * D * es, da.data_ptr() + 2 * H * D * es)
        ws = torch.empty(2 * B * H * Lq, dtype=torch.float32, device=a.device)
        _check(lib().trx_attention_bwd_ws(c(ptrs[0]), c(ptrs[1]), c(ptrs[2]), _p(m) if mode != MASK_NONE else None, mode,
                                          1 if causal else 0, B, H, Lq, Lk, ldq, ldk, float(scale), BF16, p, seed, _p(out), _p(dout),
                                          _p(lse), c(ptrs[3]), c(ptrs[4]), c(ptrs[5]), _p(ws), _stream(a)))
        return da, dkv, None, None, None, None, None


def _packed_ok(t):
    return t.is_cuda and t.dtype == torch.bfloat16 and t.shape[-1] == 64


def attention_qkv(qkv, mask=None, causal=False, scale=None, dropout_p=0.0, seed=None):
    """self-attention on a packed projection qkv [B, L, 3, H, 64] (one GEMM instead of three) -> [B, L, H*64]"""
    if scale is None:
        scale = 1.0 / math.sqrt(qkv.shape[-1])
    if dropout_p > 0 and seed is None:
        seed = new_seed()
    if qkv.dtype == torch.float16:
        return attention_qkv(qkv.to(torch.bfloat16), mask=mask, causal=causal, scale=scale,
                             dropout_p=dropout_p, seed=seed).to(torch.float16)
    if _packed_ok(qkv):
        return _AttentionPacked.apply(qkv, None, mask, causal, scale, float(dropout_p), 0 if seed is None else seed)
    q, k, v = qkv.unbind(dim=2)
    return attention(q, k, v, mask=mask, causal=causal, scale=scale, dropout_p=dropout_p, seed=seed)


def attention_q_kv(q, kv, mask=None, causal=False, scale=None, dropout_p=0.0, seed=None):
    """cross-attention: q [B, Lq, H, 64] and a packed key/value projection kv [B, Lk, 2, H, 64]"""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    if dropout_p > 0 and seed is None:
        seed = new_seed()
    if q.dtype == torch.float16:
        return attention_q_kv(q.to(torch.bfloat16), kv.to(torch.bfloat16), mask=mask, causal=causal, scale=scale,
                              dropout_p=dropout_p, seed=seed).to(torch.float16)
    if _packed_ok(q) and kv.dtype == q.dtype:
        return _AttentionPacked.apply(q, kv, mask, causal, scale, float(dropout_p), 0 if seed is None else seed)
    k, v = kv.unbind(dim=2)
    return attention(q, k, v, mask=mask, causal=causal, scale=scale, dropout_p=dropout_p, seed=seed)


# ---- Linear layers: the weight gradient on a split-contraction TN GEMM ---------------------------------------------
def gemm_tn_ok(a, b, grouped=False):
    """can trx_gemm_tn_bf16 take dW = a^T b (a [M, N], b [M, K])?  grouped: trx_gemm_tn_grouped (N: any multiple of 8)"""
    M, N = a.shape
    K = b.shape[1]
    ok = (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and b.shape[0] == M and M >= 64
          and N % (8 if grouped else 256) == 0 and K % 256 == 0 and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0
          and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and "TRX_NN_NO_GEMM" not in os.environ)
    if ok and grouped:
        # the limits of the C side (gemm_tn.hip: tn_problem_ok), so that a layer the grouped launch would refuse is never
        # deferred -- a refusal only shows when the whole list is planned, after autograd was handed None for every deferred
        # weight: 32-bit byte offsets over M + 256 rows of either operand, at most 4,096 tiles of 256 x 256 per problem (the
        # fp32 dW it writes is a fresh contiguous [N, K]: its alignment and stride hold by construction)
        ok = ((M + 256) * a.stride(0) * 2 < (1 << 32) and (M + 256) * b.stride(0) * 2 < (1 << 32)
              and ((N + 255) // 256) * (K // 256) <= 4096)
    return ok


_tn_ws = {}


def gemm_tn(a, b, colsum=False, out_dtype=torch.bfloat16):
    """a [M, N]^T . b [M, K] -> [N, K]: dW = dY^T X; with colsum=True also a.sum(0) (db), from the same pass.
    out_dtype float32: the fp32 sums are returned as they are (gradients of fp32 parameters)"""
    M, N = a.shape
    K = b.shape[1]
    nbytes = _tn_ws.get((M, N, K))
    if nbytes is None:
        nbytes = _tn_ws[(M, N, K)] = lib().trx_gemm_tn_ws_bytes(M, N, K)
    if nbytes < 0:
        raise TrxNNError("trx_gemm_tn_bf16 does not take M %d N %d K %d" % (M, N, K))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
    out = torch.empty((N, K), dtype=out_dtype, device=a.device)
    cs = torch.empty(N, dtype=out_dtype, device=a.device) if colsum else None
    _check(lib().trx_gemm_tn_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(ws), _p(out), K, _p(cs),
                                  1 if out_dtype == torch.float32 else 0, M, N, K, _stream(a)))
    return (out, cs) if colsum else out


class _TnProblem(ctypes.Structure):      # include/trx_nn.h: trx_tn_problem
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("colsum", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int)]


def gemm_tn_grouped(problems):
    """dW_i = a_i^T b_i (and db_i = a_i.sum(0)) for a LIST of (a [M, N] bf16, b [M, K] bf16, dw [N, K] fp32, db [N] fp32 or
    None) in ONE persistent launch (trx_gemm_tn_grouped_*): nothing is split over workgroups, nothing is reduced -- every
    tile's sums go straight into dw.  Stream-ordered; the plan travels through pinned memory the library owns."""
    n = len(problems)
    if n == 0:
        return
    # the table in one pack (90 problems x 10 fields set one attribute at a time were 0.35 ms of a step's host time)
    flat = []
    for (a, b, dw, db) in problems:
        flat += (a.data_ptr(), b.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else 0,
                 a.shape[0], a.shape[1], b.shape[1], a.stride(0), b.stride(0), dw.stride(0))
    arr = (_TnProblem * n).from_buffer_copy(struct.pack("<" + "4Q6i" * n, *flat))
    nbytes = lib().trx_gemm_tn_grouped_block_bytes(ctypes.addressof(arr), n)
    if nbytes < 0:
        raise TrxNNError("trx_gemm_tn_grouped: a problem this path does not take (see gemm_tn_ok)")
    dev = torch.empty(nbytes, dtype=torch.uint8, device=problems[0][0].device)
    _check(lib().trx_gemm_tn_grouped(ctypes.addressof(arr), n, dev.data_ptr(), nbytes, _stream(problems[0][0])))


_deferred = None      # a list while a deferred_wgrad() block is open: the weight gradients of the backward pass inside it


class deferred_wgrad:
    """`with ops.deferred_wgrad(): loss.backward()` -- the Linear layers' weight (and bias) gradients of this backward pass
    are computed at the END of the block by one grouped launch (gemm_tn_grouped) instead of one split-contraction call per
    layer -- and, since round 5, the add+LayerNorm calls' column sums (dgamma, dbeta, the fused bias gradient) by one launch
    instead of a second stage per call (_flush_deferred_ln) -- and are put into (or added to) the parameters' .grad directly: their backward nodes hand autograd no gradient
    for them, so per-parameter hooks do not fire (DistributedDataParallel's reducer among them; main.py reduces the
    gradients itself after the backward pass).  Outside such a block -- torch.autograd.grad -- every layer makes its own
    call as before.  (Inside a stream capture the library keeps the plan's pinned block for the life of the process.)"""

    def __enter__(self):
        global _deferred, _deferred_bytes, _deferred_ln
        self.prev, _deferred = (_deferred, _deferred_bytes, _deferred_ln), []
        _deferred_bytes, _deferred_ln = 0, []
        return self

    def __exit__(self, et, ev, tb):
        global _deferred, _deferred_bytes, _deferred_ln
        pending, pending_ln, (_deferred, _deferred_bytes, _deferred_ln) = _deferred, _deferred_ln, self.prev
        if et is None and pending:
            _flush_deferred(pending)
        if et is None and pending_ln:
            _flush_deferred_ln(pending_ln)
        return False


# dY and X of a deferred layer stay alive until its problem has run (the per-call path released them layer by layer), plus an
# fp32 dW buffer each.  TRX_NN_WGRAD_BUDGET_MB bounds what is held: past it the pending problems are launched as a group of
# their own in the middle of the backward pass.  Off by default (0), because it was measured (tools/r05/fifth.sh, B32 x L512 x
# T160, same box): the step's peak allocation is 24.2 GiB with and without a 2 GiB budget -- the peak is the forward's
# activations at the start of the backward pass, which the held operands never exceed -- while the step goes 20.8 -> 23.2 ms
# (six launches of ~15 problems fill the chip worse than one of 90); 4 GiB: 21.1 ms.
_DEFERRED_BUDGET = int(os.environ.get("TRX_NN_WGRAD_BUDGET_MB", "0")) << 20
_deferred_bytes = 0


def _flush_deferred(pending):
    gemm_tn_grouped([(dy, x, dw, db) for (dy, x, dw, db, _) in pending])
    for (_, _, dw, db, slots) in pending:
        for (param, is_bias, lo, hi) in slots:
            g = db if is_bias else dw
            if lo != 0 or hi != g.shape[0]:          # a packed projection: this parameter's row block
                g = g[lo:hi]
            if param.grad is None:
                param.grad = g
            else:
                param.grad.add_(g)


# the add+LayerNorm backward calls of the pass whose second stage (the column sums: dgamma, dbeta, the fused bias gradient) is
# still to run: (ws, nblk, cols, (gamma, beta, bias)).  One launch finishes all of them (trx_add_layernorm_bwd_reduce_many):
# a B32 x L512 x T160 step makes 42 such calls, each followed by a 5 us second stage that nothing needs before the optimizer.
_deferred_ln = None


class _LnReduceItem(ctypes.Structure):      # include/trx_nn.h: trx_ln_reduce_item
    _fields_ = [("ws", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p), ("dxbias", ctypes.c_void_p),
                ("nblk", ctypes.c_int), ("reserved", ctypes.c_int)]


def _ln_deferrable(params, needs):
    """gamma and beta (and the fused bias) are fp32 leaves that want a gradient and carry no hooks: their .grad can be assigned
    behind autograd's back.  needs = ctx.needs_input_grad of (x, res, gamma, beta, eps, p, seed, dual, bias)"""
    gamma, beta, bias = params
    if not (needs[2] and needs[3]) or (bias is not None and not needs[8]):
        return False
    ps = [t for t in params if t is not None]
    return all(t.is_leaf and t.dtype == torch.float32 for t in ps) and not _has_grad_hooks(ps)


def _flush_deferred_ln(pending):
    """One launch per (device, cols) group finishes the group's calls.  Grouped by DEVICE too: a model with LayerNorms on more than
    one GPU must not put another device's partials into one launch, and each launch runs under its own device and on that
    device's current stream.  A refusal of a grouped launch does not lose the step's gradients: the group's calls are launched one by one
    (the same HIP kernel), every call that succeeds gets its gradients, and the first error is raised when all groups are through."""
    groups = {}
    for item in pending:
        groups.setdefault((item[0].device, item[2]), []).append(item)
    first_error = None
    for (dev, cols), items in groups.items():
        n = len(items)
        with torch.cuda.device(dev):
            sums = torch.empty((n, 3, cols), dtype=torch.float32, device=dev)      # [call][dgamma | dbeta | dbias][cols]
            arr = (_LnReduceItem * n)()
            for i, (ws, nblk, _, (gamma, beta, bias)) in enumerate(items):
                base = sums[i].data_ptr()
                arr[i].ws, arr[i].dgamma, arr[i].dbeta = ws.data_ptr(), base, base + 4 * cols
                arr[i].dxbias = base + 8 * cols if bias is not None else None
                arr[i].nblk = nblk
            try:
                _check(lib().trx_add_layernorm_bwd_reduce_many(ctypes.addressof(arr), n, cols, _stream(items[0][0])))
            except Exception:
                # the grouped launch was refused: one launch per call instead (the same kernel, the same sums), so that a single bad
                # item costs its own gradients and an error, not every LayerNorm's of the step
                failed = None
                for i in range(n):
                    one = (_LnReduceItem * 1)(arr[i])
                    try:
                        _check(lib().trx_add_layernorm_bwd_reduce_many(ctypes.addressof(one), 1, cols, _stream(items[i][0])))
                    except Exception as e:      # noqa: PERF203
                        failed = e
                        sums[i].zero_()
                if failed is not None:
                    first_error = first_error or failed
        for i, (_, _, _, (gamma, beta, bias)) in enumerate(items):
            for param, g in ((gamma, sums[i, 0]), (beta, sums[i, 1]), (bias, sums[i, 2])):
                if param is None:
                    continue
                if param.grad is None:
                    param.grad = g
                else:
                    param.grad.add_(g)
    if first_error is not None:
        raise first_error


def _has_grad_hooks(params):
    """a parameter with a tensor hook or a post-accumulate hook wants its gradient from autograd: the deferred path assigns
    .grad directly and no hook would fire, so such a layer takes the per-call path"""
    for p in params:
        if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return True
    return False


def backward(loss):
    """loss.backward() with the weight gradients of the Linear layers deferred into one grouped launch"""
    if loss.is_cuda and os.environ.get("TRX_NN_WGRAD", "grouped") != "percall":      # (the switch: same-box A/B)
        with deferred_wgrad():
            loss.backward()
    else:
        loss.backward()


class WeightShadows:
    """bf16 copies of the Linear parameters a model feeds to `linear`, refreshed by ONE multi-tensor copy per forward.

    Autocast casts every fp32 weight and bias with its own kernel at every step (~200 launches of a few microseconds,
    and as many on the host, which bounds the step on real, ragged batches: DESIGN.md section 8).  A group = the weights
    (and biases) one `linear` call concatenates along the outputs -- (query, key, value) of a self-attention, (key, value)
    of a cross-attention, or a single Linear -- stored back to back in one bf16 buffer, so that the packed projection
    needs no torch.cat either.  Groups are registered the first time a forward uses them; `refresh()` at the start of
    every forward re-copies all of them (parameters change between forwards, not inside one)."""

    def __init__(self):
        self.groups = {}          # (tuple(id(weight) ...), has biases) -> (weights, biases, w16, b16)
        self._dst, self._src = [], []
        self.generation = 0       # moves when the parameters behind the shadows are known to have changed (mark_stale)

    def mark_stale(self):
        """the trainer calls this after an optimizer step / load_state_dict: a backward whose forward ran before it would
        read the NEXT forward's values out of the shared buffers, and says so instead (`_LinearWgrad.backward`).  Two
        forwards before their backwards (no step in between) keep one generation and stay legal."""
        self.generation += 1

    # a copy or a pickle of the model starts with an empty registry (views into packed buffers do not survive either)
    def __deepcopy__(self, memo):
        return WeightShadows()

    def __reduce__(self):
        return (WeightShadows, ())

    def refresh(self):
        """re-copy every group.  (Skipping the copy when no parameter's autograd version moved would be cheaper, but fused
        optimizers -- AdamW(fused=True), what the trainer uses -- update parameters without moving it.)"""
        if not self.groups:
            return
        dev = self._dst[0].device
        if any(p.device != dev for p in self._src):       # the model moved: rebuild lazily
            self.groups, self._dst, self._src = {}, [], []
            return
        with torch.no_grad():
            torch._foreach_copy_(self._dst, [p.detach() for p in self._src])

    def lookup(self, weights, biases):
        key = (tuple(id(w) for w in weights), biases is not None)
        g = self.groups.get(key)
        if g is None:
            with torch.no_grad():
                outs = [w.shape[0] for w in weights]
                w16 = torch.empty((sum(outs), weights[0].shape[1]), dtype=torch.bfloat16, device=weights[0].device)
                b16 = torch.empty(sum(outs), dtype=torch.bfloat16, device=weights[0].device) if biases is not None else None
                lo = 0
                for i, w in enumerate(weights):
                    wv = w16[lo:lo + outs[i]]
                    wv.copy_(w.detach())
                    self._dst.append(wv); self._src.append(w)
                    if biases is not None:
                        bv = b16[lo:lo + outs[i]]
                        bv.copy_(biases[i].detach())
                        self._dst.append(bv); self._src.append(biases[i])
                    lo += outs[i]
            g = self.groups[key] = (tuple(weights), None if biases is None else tuple(biases), w16, b16)
        return g[2], g[3]


_shadows = None


class use_shadows:
    """`with ops.use_shadows(registry):` around a forward pass: `linear` takes its bf16 operands from the registry"""

    def __init__(self, registry):
        self.registry = registry

    def __enter__(self):
        global _shadows
        self.prev, _shadows = _shadows, self.registry
        if self.registry is not None:
            self.registry.refresh()
        return self.registry

    def __exit__(self, *exc):
        global _shadows
        _shadows = self.prev
        return False


class _LinearWgrad(torch.autograd.Function):
    """y = x [W_1; W_2; ...]^T (+ [b_1; b_2; ...]) with the library GEMM; in the backward the weight gradient dY^T X runs on
    trx_gemm_tn_bf16 (the library contracts the 16,384 token rows inside 9 .. 36 workgroups: 0.48 PFLOP/s), dx stays a
    library call.  apply(x, n, W_1 .. W_n[, b_1 .. b_n]): several Linear layers over the same input (the query / key /
    value projections) are ONE product; their gradients are row blocks of the one dW."""

    @staticmethod
    def forward(ctx, x, n, *params):
        weights = params[:n]
        biases = params[n:] if len(params) > n else None
        # the parameters arrive in their own precision (fp32 under autocast): the bf16 operands come from the model's
        # WeightShadows when a forward has installed them, from casts otherwise; the backward hands back gradients in the
        # parameters' precision straight from the fp32 sums (no bf16 rounding, no cast kernels)
        bf = torch.bfloat16
        if _shadows is not None and weights[0].dtype == torch.float32:
            w16, b16 = _shadows.lookup(weights, biases)
            ctx.shadow = (_shadows, _shadows.generation)
        else:
            ctx.shadow = None
            w16 = weights[0].to(bf) if n == 1 else torch.cat([w.to(bf) for w in weights])
            b16 = None if biases is None else (biases[0].to(bf) if n == 1 else torch.cat([b.to(bf) for b in biases]))
        # w16 rides on ctx, not in save_for_backward: a shadow is re-copied (with the same values) by the next forward, and
        # two forwards before one backward must not trip autograd's in-place check -- the semantics autocast's own cached
        # casts have (a parameter changed between a forward and its backward goes unnoticed there as well)
        ctx.save_for_backward(x)
        ctx.w16 = w16
        ctx.params = params
        ctx.n, ctx.has_bias = n, biases is not None
        ctx.outs = [w.shape[0] for w in weights]
        ctx.wdtype = weights[0].dtype
        # autocast off: under fp16 autocast (--precision 16-mixed) the library call would be re-cast to fp16 and hand the
        # backward an fp16 gradient for bf16 operands; x is bf16 here by construction (see `linear`)
        was = torch.is_autocast_enabled("cuda")          # (the flag itself: entering a torch.autocast context costs ~6 us, 90 times a step)
        torch.set_autocast_enabled("cuda", False)
        try:
            return torch.nn.functional.linear(x, w16, b16)
        finally:
            torch.set_autocast_enabled("cuda", was)

    @staticmethod
    def backward(ctx, dy):
        (x,), w16 = ctx.saved_tensors, ctx.w16
        if ctx.shadow is not None and ctx.shadow[0].generation != ctx.shadow[1]:
            raise RuntimeError("linear backward: the parameters were updated (optimizer step / load_state_dict) between this "
                               "forward and its backward; the shared bf16 weight copies now hold the new values")
        n = ctx.n
        dx = dw = db = None
        dy2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16)      # a no-op except behind an fp16-autocast consumer
        x2 = x.reshape(-1, x.shape[-1])
        if ctx.needs_input_grad[0]:
            dx = torch.matmul(dy2, w16).view(x.shape)
        need_w = any(ctx.needs_input_grad[2:2 + n])
        want_db = ctx.has_bias and any(ctx.needs_input_grad[2 + n:])
        if (need_w and _deferred is not None and ctx.wdtype == torch.float32 and gemm_tn_ok(dy2, x2, grouped=True)
                and not _has_grad_hooks(ctx.params)):
            # deferred_wgrad(): this layer's problem joins the pass's one grouped launch; autograd gets no gradient for the
            # parameters here -- the end of the block assigns them
            ntot = dy2.shape[1]
            dwbuf = torch.empty((ntot, x2.shape[1]), dtype=torch.float32, device=dy2.device)
            dbbuf = torch.empty(ntot, dtype=torch.float32, device=dy2.device) if want_db else None
            slots, lo = [], 0
            for j, o in enumerate(ctx.outs):
                if ctx.needs_input_grad[2 + j]:
                    slots.append((ctx.params[j], False, lo, lo + o))
                if want_db and ctx.needs_input_grad[2 + n + j]:
                    slots.append((ctx.params[n + j], True, lo, lo + o))
                lo += o
            _deferred.append((dy2, x2, dwbuf, dbbuf, slots))
            global _deferred_bytes
            _deferred_bytes += dy2.numel() * 2 + x2.numel() * 2 + dwbuf.numel() * 4
            if _DEFERRED_BUDGET and _deferred_bytes > _DEFERRED_BUDGET:
                pending = list(_deferred)
                del _deferred[:]
                _deferred_bytes = 0
                _flush_deferred(pending)
            return (dx, None) + (None,) * len(ctx.params)
        if need_w:
            od = torch.float32 if ctx.wdtype == torch.float32 else torch.bfloat16
            if gemm_tn_ok(dy2, x2):        # any token count >= 64: the kernel zero-fills the rows past the end of its last step
                if want_db:
                    dw, db = gemm_tn(dy2, x2, colsum=True, out_dtype=od)
                else:
                    dw = gemm_tn(dy2, x2, out_dtype=od)
            else:
                dw = torch.matmul(dy2.t(), x2).to(ctx.wdtype)
        if want_db and db is None:
            db = dy2.sum(dim=0)
        if db is not None:
            db = db.to(ctx.wdtype)
        grads, lo = [], 0
        for o in ctx.outs:                                         # row blocks of the one dW / db
            grads.append(dw[lo:lo + o] if dw is not None else None)
            lo += o
        if ctx.has_bias:
            lo = 0
            for o in ctx.outs:
                grads.append(db[lo:lo + o] if db is not None else None)
                lo += o
        return (dx, None, *grads)


_packed_cache = None


class packed_projections:
    """with ops.packed_projections(): ... -- for the duration, `linear_multi` under no_grad + bf16 autocast keeps the packed bf16
    weights it builds (one per group of parameters) instead of rebuilding them at every call.  The caller vouches that the
    parameters do not change inside the block (dense.encode: one inference pass over a corpus).  Deliberately not keyed on
    the parameters' autograd version: fused optimizers update parameters without moving it (WeightShadows.refresh)."""

    def __enter__(self):
        global _packed_cache
        self._outer, _packed_cache = _packed_cache, ({} if _packed_cache is None else _packed_cache)
        return self

    def __exit__(self, *exc):
        global _packed_cache
        _packed_cache = self._outer
        return False


def linear_multi(x, weights, biases=None):
    """F.linear(x, cat(weights), cat(biases)) for Linear layers that read the same input (query / key / value): one
    product, no concatenation of the fp32 parameters (see _LinearWgrad); falls back to the concatenation otherwise"""
    w0 = weights[0]
    if (x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and w0.requires_grad
            and sum(w.shape[0] for w in weights) % 256 == 0 and w0.shape[1] % 256 == 0):
        return _LinearWgrad.apply(x, len(weights), *weights, *(biases if biases is not None else ()))
    if (_packed_cache is not None and len(weights) > 1 and x.is_cuda and not torch.is_grad_enabled()
            and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        # inside `packed_projections()` (dense.encode: ~100 batches through the same 12 layers, the parameters fixed for the
        # duration): the packed bf16 weight of a layer is made once; concatenating the fp32 parameters and casting the result
        # cost four launches and 20 us per layer and batch, 4 % of an encoder forward at 256 x ~74 tokens
        key = tuple(id(t) for t in weights) + tuple(id(t) for t in (biases or ()))
        hit = _packed_cache.get(key)
        if hit is None:
            w = torch.cat([t.detach() for t in weights]).to(torch.bfloat16)
            b = None if biases is None else torch.cat([t.detach() for t in biases]).to(torch.bfloat16)
            hit = _packed_cache[key] = (w, b, weights, biases)      # (the parameters themselves: their ids stay theirs while the cache lives)
        return torch.nn.functional.linear(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), hit[0], hit[1])
    w = w0 if len(weights) == 1 else torch.cat(list(weights))
    b = None if biases is None else (biases[0] if len(biases) == 1 else torch.cat(list(biases)))
    return linear(x, w, b)


def linear(x, weight, bias=None):
    """torch.nn.functional.linear; for bf16 activations on the GPU (autocast training) the weight gradient is routed
    to the split-contraction TN GEMM when its shape qualifies"""
    if (x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and weight.requires_grad
            and weight.shape[0] % 8 == 0 and weight.shape[1] % 256 == 0):      # (N % 256 != 0, the vocabulary projection: the grouped launch takes it, a single call hands it to the library)
        return _LinearWgrad.apply(x, 1, weight, *((bias,) if bias is not None else ()))
    if (x.is_cuda and x.dtype == torch.bfloat16 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.float16):
        # fp16 autocast (--precision 16-mixed) over the bf16 stream the LayerNorm kernel writes: keep the product in bf16
        # (autocast would re-cast x to fp16 and every HIP op downstream would convert it back)
        with torch.autocast("cuda", enabled=False):
            bf = torch.bfloat16
            return torch.nn.functional.linear(x, weight.to(bf), bias.to(bf) if bias is not None else None)
    return torch.nn.functional.linear(x, weight, bias)


# ---- torch.ops.trx.*: the same launchers behind torch.library registrations ("exposes the kernel as
# torch.ops.trx.attention_fwd / bwd backed by extern "C" launchers taking raw device pointers + hipStream_t").  The model
# calls the autograd Functions above directly (no dispatcher hop on a path of ~100 calls per step); a caller that wants
# dispatcher-visible ops -- a module swap inside the reference, torch.compile -- uses these.  GPU only: like everything in
# this file they fail on CPU tensors instead of falling back.
_trx_lib = torch.library.Library("trx", "DEF")
_trx_lib.define("attention_fwd(Tensor q, Tensor k, Tensor v, Tensor? mask, bool causal, float scale, float p, int seed) -> (Tensor, Tensor)")
_trx_lib.define("attention_bwd(Tensor q, Tensor k, Tensor v, Tensor? mask, bool causal, float scale, float p, int seed, "
                "Tensor out, Tensor dout, Tensor lse) -> (Tensor, Tensor, Tensor)")
_trx_lib.define("add_layernorm_fwd(Tensor x, Tensor? res, Tensor gamma, Tensor beta, float eps, float p, int seed) -> (Tensor, Tensor, Tensor)")
_trx_lib.define("add_layernorm_bwd(Tensor dy, Tensor x, Tensor? res, Tensor gamma, Tensor mean, Tensor rstd, float p, int seed) "
                "-> (Tensor, Tensor, Tensor, Tensor)")


def _op_attention_fwd(q, k, v, mask, causal, scale, p, seed):
    out, lse, _, _ = _attention_fwd_launch(q.contiguous(), k.contiguous(), v.contiguous(), mask, causal, scale, p, seed, True)
    return out, lse


def _op_attention_bwd(q, k, v, mask, causal, scale, p, seed, out, dout, lse):
    m = mask.float().contiguous() if mask is not None else None
    mode = MASK_NONE if m is None else (MASK_KEY if m.dim() == 2 else MASK_FULL)
    return _attention_bwd_launch(q.contiguous(), k.contiguous(), v.contiguous(), m, mode, causal, scale, p, seed, out, dout, lse)


def _op_add_ln_fwd(x, res, gamma, beta, eps, p, seed):
    y, mean, rstd, _, _, _ = _add_ln_fwd_launch(x, res, gamma, beta, eps, p, seed, True)
    return y, mean, rstd


def _op_add_ln_bwd(dy, x, res, gamma, mean, rstd, p, seed):
    dz, dx, dg, db = _add_ln_bwd_launch(dy, x.contiguous(), res.contiguous() if res is not None else None,
                                        gamma.float().contiguous(), mean, rstd, p, seed)
    return (dx.clone() if dx is dz else dx), dz, dg, db      # outputs of a custom op must not alias each other


_trx_lib.impl("attention_fwd", _op_attention_fwd, "CUDA")
_trx_lib.impl("attention_bwd", _op_attention_bwd, "CUDA")
_trx_lib.impl("add_layernorm_fwd", _op_add_ln_fwd, "CUDA")
_trx_lib.impl("add_layernorm_bwd", _op_add_ln_bwd, "CUDA")


@torch.library.register_fake("trx::attention_fwd")
def _(q, k, v, mask, causal, scale, p, seed):
    B, Lq, H, D = q.shape
    return q.new_empty((B, Lq, H * D)), q.new_empty((B, H, Lq), dtype=torch.float32)


@torch.library.register_fake("trx::attention_bwd")
def _(q, k, v, mask, causal, scale, p, seed, out, dout, lse):
    return torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)


@torch.library.register_fake("trx::add_layernorm_fwd")
def _(x, res, gamma, beta, eps, p, seed):
    rows = x.numel() // x.shape[-1]
    return torch.empty_like(x), x.new_empty(rows, dtype=torch.float32), x.new_empty(rows, dtype=torch.float32)


@torch.library.register_fake("trx::add_layernorm_bwd")
def _(dy, x, res, gamma, mean, rstd, p, seed):
    c = x.shape[-1]
    return torch.empty_like(x), torch.empty_like(x), x.new_empty(c, dtype=torch.float32), x.new_empty(c, dtype=torch.float32)


def _attn_setup(ctx, inputs, output):
    q, k, v, mask, causal, scale, p, seed = inputs
    out, lse = output
    ctx.save_for_backward(q, k, v, mask if mask is not None else q.new_empty(0), out, lse)
    ctx.has_mask = mask is not None
    ctx.cfg = (causal, scale, p, seed)


def _attn_backward(ctx, dout, dlse):
    q, k, v, mask, out, lse = ctx.saved_tensors
    causal, scale, p, seed = ctx.cfg
    dq, dk, dv = torch.ops.trx.attention_bwd(q, k, v, mask if ctx.has_mask else None, causal, scale, p, seed, out, dout.contiguous(), lse)
    return dq, dk, dv, None, None, None, None, None


def _ln_setup(ctx, inputs, output):
    x, res, gamma, beta, eps, p, seed = inputs
    y, mean, rstd = output
    ctx.save_for_backward(x, res if res is not None else x.new_empty(0), gamma, mean, rstd)
    ctx.has_res = res is not None
    ctx.drop = (p, seed)


def _ln_backward(ctx, dy, dmean, drstd):
    x, res, gamma, mean, rstd = ctx.saved_tensors
    p, seed = ctx.drop
    dx, dz, dg, db = torch.ops.trx.add_layernorm_bwd(dy.contiguous(), x, res if ctx.has_res else None, gamma, mean, rstd, p, seed)
    return dx, (dz if ctx.has_res else None), dg.to(gamma.dtype), db.to(gamma.dtype), None, None, None


torch.library.register_autograd("trx::attention_fwd", _attn_backward, setup_context=_attn_setup)
torch.library.register_autograd("trx::add_layernorm_fwd", _ln_backward, setup_context=_ln_setup)
