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
                ops.backward(loss)          # what the trainer calls (main.py): the weight gradients as one grouped launch
                opt.step()
            else:
                ops.backward(scaler.scale(loss))
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


def live_problem(P, N, Lp=128, Lq=128, vocab=31090, seed=0):
    """BASELINE.json configs[4] at the size of scripts/train_RetroSyn_tf.sh: a USPTO-50K-sized training split (N ~ 40 k
    product SMILES) against a text corpus of P passages, pre-tokenised (tokenizers are out of scope): passage lengths uniform
    in [16, Lp], query lengths in [16, Lq], every 7th passage a duplicate text (the de-duplication has work to do), the
    gold passage of 3 in 4 samples present in the corpus.  Seeded; the same tensors on every rank."""
    g = torch.Generator().manual_seed(seed)
    plen = torch.randint(16, Lp + 1, (P,), generator=g)
    pids = torch.randint(1000, vocab, (P, Lp), generator=g, dtype=torch.int32)
    dup = torch.arange(7, P, 7)
    pids[dup], plen[dup] = pids[dup - 2], plen[dup - 2]
    corpus = {"passage_ids": pids, "passage_len": plen, "marker_ids": torch.tensor([[1006, 1014 + j, 1007] for j in range(3)]),
              "cls_id": 101, "sep_id": 102, "pad_id": 0, "mask_id": 103}
    qlen = torch.randint(16, Lq + 1, (N,), generator=g)
    qids = torch.randint(1000, vocab, (N, Lq), generator=g)
    gold = torch.randint(0, P, (N,), generator=g)
    gold[::4] = -1
    return corpus, qids, qlen, gold


def live_bench(dev, P=204800, N=40000, k=10, rank=0, world=1, encoder=None, max_length=512, train_step_ms=None,
               return_tensors=False):
    """One refresh of the on-the-fly retrieval (textreact_amd/live.py, main.py --live_every) at the scripts' size, stage by
    stage: BERT-base DenseEncoder ([CLS] embeddings, bf16 autocast, the HIP attention / add+LayerNorm kernels) over this
    rank's passages -> flat index add -> this rank's 1/G of the queries + one all-gather -> row-sharded exact top-k ->
    per-epoch assembly of the encoder inputs (select_neighbors + assemble_inputs + apply_mlm with train_RetroSyn_tf.sh's
    options).  BASELINE.md C4: "retrieval refresh time per epoch" beside the epoch's train time."""
    import time
    from textreact_amd import dense, live
    from textreact_amd.predictor.model import Config
    cd, qids, qlen, gold = live_problem(P, N)
    corpus = live.LiveCorpus(cd, dev)
    if encoder is None:
        torch.manual_seed(0)
        encoder = dense.DenseEncoder(Config(vocab_size=31090)).to(dev).eval()
    ms = {}
    nn, emb_q, emb_p = live.refresh_neighbors(encoder, encoder, corpus, qids, qlen, k, rank, world, return_embeddings=True,
                                              timings={})          # warm-up: allocator, library heuristics, index workspaces
    del nn, emb_q, emb_p
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nn, emb_q, emb_p = live.refresh_neighbors(encoder, encoder, corpus, qids, qlen, k, rank, world, return_embeddings=True, timings=ms)
    torch.cuda.synchronize()
    refresh_ms = (time.perf_counter() - t0) * 1e3
    # the epoch's inputs, as main.py's LiveData.inputs builds them (blocks of 32,768 samples)
    gen = torch.Generator().manual_seed(1)
    t0 = time.perf_counter()
    widest = 0
    for b0 in range(0, N, 32768):
        r = slice(b0, b0 + 32768)
        sel = live.select_neighbors(nn[r], gold[r].to(dev), corpus, True, use_gold_neighbor=True, max_num_neighbors=10, num_neighbors=3,
                                    random_neighbor_ratio=0.2, generator=gen)
        ids, mask, lens = live.assemble_inputs(qids[r], qlen[r], sel, corpus, max_length)
        ids, pos, labels = live.apply_mlm(ids, lens, 0.15, corpus.mask_id, gen)
        widest = max(widest, ids.shape[1])
    torch.cuda.synchronize()
    assemble_ms = (time.perf_counter() - t0) * 1e3
    uncert = ms.pop("uncertified_queries", None)
    rescored = ms.pop("rescored_queries", None)
    rescanned = ms.pop("rescanned_queries", None)
    # the search stage alone on embeddings with the spread of a TRAINED retriever: with random-init weights every [CLS]
    # embedding is nearly the same vector (cosine of two passages > 0.99), the top-10 scores of a query lie within the
    # certificate's a-priori rounding bound of each other, and every query takes the exact fp64 re-scan -- a worst case that
    # says nothing about a real refresh.  Same shapes, Gaussian embeddings (BASELINE.json configs[1]'s data):
    from textreact_amd import faiss_compat
    from textreact_amd.sharded import ShardedFlatIndex, shard_bounds
    g = torch.Generator(device=dev); g.manual_seed(5)
    lo, hi = shard_bounds(P, world, rank)
    gp = torch.randn((P, 768), generator=g, device=dev).bfloat16()[lo:hi].contiguous()
    gq = torch.randn((N, 768), generator=g, device=dev).bfloat16()
    gi = ShardedFlatIndex(768, 0, local_index=faiss_compat.IndexFlatIP(768, device=dev.index or 0))
    gi.add_shard(gp, lo, P)
    gi.search(gq, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gi.search(gq, k)
    torch.cuda.synchronize()
    search_gauss_ms = (time.perf_counter() - t0) * 1e3
    gauss_uncert = gi.local.last_stats()["n_uncertified"]
    del gi, gp, gq
    cosine = float(torch.nn.functional.cosine_similarity(emb_p[:2048:2].float(), emb_p[1:2048:2].float()).mean())
    steps_per_epoch = -(-N // (32 * world))
    row = {"kernel": "live_refresh", "what": "BASELINE.json configs[4]: on-the-fly retrieval refresh at the size of train_RetroSyn_tf.sh",
           "encoder": "BERT-base DenseEncoder (12 layers, hidden 768, vocab 31090), bf16 autocast, random init",
           "passages": P, "passage_tokens": "16-128 + [CLS] [SEP]", "queries": N, "query_tokens": "16-128 + [CLS] [SEP]", "k": k,
           "ranks": world, "ms": {**{k_: round(v, 2) for k_, v in ms.items()}, "assemble_epoch_inputs": round(assemble_ms, 2)},
           "refresh_ms": round(refresh_ms, 2), "refresh_plus_assemble_ms": round(refresh_ms + assemble_ms, 2),
           "uncertified_queries": uncert, "rescored_queries": rescored, "rescanned_queries": rescanned, "mean_cosine_of_passage_pairs": round(cosine, 4),
           "search_ms_on_gaussian_embeddings_of_the_same_shape": round(search_gauss_ms, 2), "uncertified_on_gaussian": gauss_uncert,
           "assembled_width": widest,
           "passages_per_s_encode": (P / world) / (ms["encode_passages"] * 1e-3),
           "steps_per_epoch": steps_per_epoch, "train_step_ms": train_step_ms,
           "epoch_train_ms": None if train_step_ms is None else train_step_ms * steps_per_epoch,
           "refresh_share_of_epoch": None if train_step_ms is None else (refresh_ms + assemble_ms) / (train_step_ms * steps_per_epoch + refresh_ms + assemble_ms)}
    if world > 1:
        row["transport"] = "REHEARSAL: %d ranks share one GPU over gloo (host round trips in the collectives; the stages time-slice the device): not a scaling number" % world
    return (row, nn, emb_q, emb_p) if return_tensors else row


def _live_rank(rank, world, port, P, N, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    row = live_bench(dev, P, N, rank=rank, world=world)
    if rank == 0:
        ret["row"] = row
    dist.barrier()
    dist.destroy_process_group()


def live_rows(dev, P=204800, N=40000):
    """--live: the refresh on one rank (with the train step it is to be read against), then as a 2-rank one-device rehearsal"""
    import socket
    import torch.multiprocessing as mp
    from textreact_amd.predictor.model import Config
    from textreact_amd.predictor import train
    # the epoch's train time: the step of train_RetroSyn_tf.sh's shapes (B 32 per GPU, L 512, T 160, bf16 autocast, --mlm)
    B, L, T = 32, 512, 160
    g = torch.Generator().manual_seed(0)
    batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
             "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
    labels = {"mlm_labels": torch.randint(0, 31090, (B, 76), generator=g).to(dev)}
    torch.manual_seed(0)
    p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True),
                        mlm=True).to(dev).train()
    opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss, _ = p.training_step(batch, labels)
        ops.backward(loss); opt.step(); opt.zero_grad(set_to_none=True)
    step_ms = timeit(step, iters=8, warm=4)
    del p, opt
    torch.cuda.empty_cache()
    rows = [live_bench(dev, P, N, train_step_ms=step_ms)]
    torch.cuda.empty_cache()
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_live_rank, args=(2, port, P, N, ret), nprocs=2, join=True)
    rows.append(dict(ret["row"]))
    return rows


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
            # both rooflines: the matrix pipe (4 B H Lq Lk 64 FLOP) and HBM (q, k, v read once, out written once); the
            # decoder's shapes (160 queries) sit nearer the second -- `bound` names the one that binds
            es = 2 if dtype == torch.bfloat16 else 4
            alg_bytes = float(B * H * 64 * es * (2 * Lq + 2 * Lk))
            gbs = alg_bytes / (ms * 1e-3) / 1e9
            peak_tf = 2500.0 if dtype == torch.bfloat16 else 157.3      # fp32: v_mfma_f32_16x16x4_f32 = the fp32 vector peak
            mfma_frac, hbm_frac = tf / peak_tf, gbs / 8000.0
            out.append({"kernel": "attention_fwd", "what": name, "dtype": str(dtype), "B": B, "H": H, "Lq": Lq, "Lk": Lk,
                        "ms": ms, "torch_eager_fp32_ms": ref,
                        "roofline": {"bound": "mfma" if mfma_frac >= hbm_frac else "hbm", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s",
                                     "frac": mfma_frac, "hbm_frac": hbm_frac, "hbm_achieved_GBps": gbs, "algorithmic_bytes": alg_bytes}})
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
    # the same products the way the trainer runs them (ops.backward -> trx_gemm_tn_grouped_*): ALL weight gradients of one
    # training step (B32 . L512 . T160: 48 encoder + 42 decoder Linear layers) as one persistent launch -- no split, no
    # partials, no reduction launch
    enc = [(32 * 512, 2304, 768), (32 * 512, 768, 768), (32 * 512, 3072, 768), (32 * 512, 768, 3072)] * 12
    dec = [(32 * 160, 2304, 768), (32 * 160, 768, 768), (32 * 160, 768, 768), (32 * 512, 1536, 768), (32 * 160, 768, 768), (32 * 160, 3072, 768),
           (32 * 160, 768, 3072)] * 6
    shapes = sorted(set(enc + dec))
    ops_ab = {sh: (torch.randn(sh[0], sh[1], device=dev).to(torch.bfloat16), torch.randn(sh[0], sh[2], device=dev).to(torch.bfloat16)) for sh in shapes}
    probs = [(ops_ab[sh][0], ops_ab[sh][1], torch.empty(sh[1], sh[2], device=dev), torch.empty(sh[1], device=dev)) for sh in enc + dec]
    ms = timeit(lambda: ops.gemm_tn_grouped(probs), iters=10)
    fl = sum(2.0 * m * n * k for (m, n, k) in enc + dec)
    percall = sum(timeit(lambda: ops.gemm_tn(ops_ab[sh][0], ops_ab[sh][1], colsum=True, out_dtype=torch.float32), iters=10) * (enc + dec).count(sh) for sh in shapes)
    tf = fl / (ms * 1e-3) / 1e12
    out.append({"kernel": "gemm_tn grouped (dW + db of every Linear layer of a step)", "problems": len(probs), "flop": fl, "ms": ms,
                "per_call_form_ms": percall, "roofline": {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0}})
    del probs, ops_ab
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
    elif len(sys.argv) > 1 and sys.argv[1] == "--live":       # [passages [queries]]
        for row in live_rows(torch.device("cuda", 0), *[int(a) for a in sys.argv[2:4]]):
            print(json.dumps(row), flush=True)
    else:
        main()
