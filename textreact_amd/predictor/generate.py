"""Beam-search decoding for the predictor's test step (textreact/main.py:218-233).

The reference calls Hugging Face `generate` on its EncoderDecoderModel:

    output = self.model.generate(**batch_in, num_beams=num_beams, num_return_sequences=num_beams,
                                 max_length=self.args.max_dec_length, length_penalty=0,
                                 bos_token_id=..., eos_token_id=..., pad_token_id=...,
                                 return_dict_in_generate=True, output_scores=True)

i.e. deterministic beam search (no sampling, no logits processors, `early_stopping=False`), or greedy
search when num_beams == 1.  This module restates that algorithm -- transformers 4.27.3
`GenerationMixin.beam_search` + `BeamSearchScorer.process / finalize` + `BeamHypotheses`, the version
the reference pins -- over TextReactModel with a key/value cache: the encoder runs once, the
cross-attention keys / values of every decoder layer are projected once, each step feeds one token per
beam through the decoder (attention and add+LayerNorm through the same ops as training), and the
self-attention cache is re-ordered by the surviving beams' parents.

Golden: tests/golden/generate_small.npz holds what `generate` itself returns for the reference model
(tests/golden/make_generate_golden.py); sequences agree up to the padding after the end token (4.27.3
pads with pad_token_id, the transformers 5.x that produced the golden repeats the end token;
`batch_decode(skip_special_tokens=True)`, what main.py:227 does next, erases both), scores to 1e-4.
"""
import contextlib
import os

import torch

from . import ops


class _BeamHypotheses:
    """n-best list of finished hypotheses of one batch item ([3P] generation/beam_search.py: BeamHypotheses)"""

    def __init__(self, num_beams, length_penalty, early_stopping):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                ranked = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping is True:
            return True
        if self.early_stopping is False:
            return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty
        # "never"
        if self.length_penalty > 0.0:
            raise ValueError("early_stopping='never' with a positive length penalty needs max_length here")
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


_warmup_streams = {}


def _use_graph(model, dev, graph):
    """decode through one captured HIP graph per generate call?  Default: yes (TRX_DECODE_GRAPH=0 or graph=False turns it
    off and runs the same step as eager launches)."""
    if graph is None:
        graph = os.environ.get("TRX_DECODE_GRAPH", "1") != "0"
    return bool(graph) and dev.type == "cuda"


class _DecoderState:
    """encoder output, per-layer cross-attention K / V (projected once) and the self-attention cache.

    graph=True: one decode step -- re-order the cache by the surviving beams' parents, embed the new tokens,
    all decoder layers, the LM head and log-softmax, ~150 launches of a few microseconds each and therefore
    bound by the host -- is captured into a HIP graph over static buffers and replayed per position (with beams: two
    graphs, for even and odd positions, that re-order the cache from one of two cache sets into the other; on the bf16
    path the cache is not re-ordered at all -- a table of ancestors is, and the attention kernel follows it).
    What changes from step to step lives in device tensors the graph reads: the tokens, the parents, and
    the position t (the cache row written with index_copy_, the position ids, and a key mask over the
    full-length cache that opens one more column per step, which gives the same softmax as the eager
    loop's attention over the first t + 1 cache rows: masked keys contribute exactly 0)."""

    def __init__(self, model, input_ids, attention_mask, expand, max_length, graph=False):
        from .model import additive_key_mask
        self.model = model
        dec = model.decoder.roberta
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        key = additive_key_mask(attention_mask)
        enc = model.encoder(input_ids, key, None, None, None)
        self.enc_states = enc
        # the `expand` beams of one input share its encoder states: they are the QUERY ROWS of one cross-attention
        # problem per input (q [B, expand, H, 64] against K / V [B, L, H, 64]), not `expand` copies of K / V
        self.key, self.expand = key, expand
        B, L, _ = enc.shape
        n = B * expand
        self.layers = list(dec.encoder.layer)
        H = self.layers[0].attention.heads
        self.H = H
        self.pad = dec.embeddings.pad
        self.graph = None
        dev = enc.device
        # bf16 autocast + graph: the step runs on bf16 copies of the decoder weights made once per call (inside a
        # capture autocast may not cache its casts, so the module path would cast every weight at every replay),
        # with the key / value projections packed (one GEMM, one cache tensor and one cache write per layer)
        # (fp16 autocast, the scripts' --precision 16-mixed, takes the same bf16 path: the HIP kernels store bf16 or fp32)
        self.fast = bool(graph and torch.is_autocast_enabled("cuda")
                         and torch.get_autocast_dtype("cuda") in (torch.bfloat16, torch.float16))
        # with beams the fast path never re-orders its cache: an ancestor table is re-ordered instead and the attention kernel
        # follows it (ops.attention_decode_gather; the table holds up to 256 positions)
        self.gather = self.fast and expand > 1 and max_length <= 256
        if self.fast:
            with torch.autocast("cuda", enabled=False):      # every cast of the fast path is explicit
                self._prepare_fast(enc, n, max_length)
        else:
            self.kx = [ly.crossattention.self.key(enc).view(B, L, H, 64) for ly in self.layers]
            self.vx = [ly.crossattention.self.value(enc).view(B, L, H, 64) for ly in self.layers]
            dt = self.kx[0].dtype
            self.kc = [torch.zeros((n, max_length, H, 64), dtype=dt, device=dev) for _ in self.layers]
            self.vc = [torch.zeros((n, max_length, H, 64), dtype=dt, device=dev) for _ in self.layers]
        if graph:
            self._capture(n, max_length, dev, reorder=expand > 1)

    # ---- eager step ----------------------------------------------------------------------------------------
    def _layers(self, h, self_attention):
        m, H = self.model, self.H
        n = h.shape[0]
        for li, ly in enumerate(self.layers):
            at = ly.attention
            q = at.self.query(h).view(n, 1, H, 64)
            ctx = self_attention(li, q, at.self.key(h), at.self.value(h))
            h = ops.add_layernorm(at.output.dense(ctx), h, at.output.LayerNorm.weight, at.output.LayerNorm.bias, at.eps)
            ca = ly.crossattention
            q = ca.self.query(h).view(n // self.expand, self.expand, H, 64)
            ctx = ops.attention(q, self.kx[li], self.vx[li], mask=self.key, causal=False).view(n, 1, H * 64)
            h = ops.add_layernorm(ca.output.dense(ctx), h, ca.output.LayerNorm.weight, ca.output.LayerNorm.bias, ca.eps)
            f = torch.nn.functional.gelu(ly.intermediate.dense(h))
            h = ops.add_layernorm(ly.output.dense(f), h, ly.output.LayerNorm.weight, ly.output.LayerNorm.bias, ly.eps)
        logits = m.decoder.lm_head(h)[:, -1]
        return torch.log_softmax(logits.float(), dim=-1)

    def step(self, tokens, t):
        """tokens [n] = the tokens at position t; returns log-probabilities [n, vocab] of position t + 1"""
        if self.graph is not None:
            self.g_tok.copy_(tokens)
            self.g_t.fill_(t)
            which = t % len(self.graph)           # with beams: even / odd positions alternate between two cache sets
            self.graph[which].replay()
            return self.g_logp[which]
        H = self.H
        ids = tokens[:, None]
        # RoBERTa positions with a cache: (1 + past length) for real tokens, the padding index for padding
        pos = torch.where(ids.ne(self.pad), torch.full_like(ids, t + 1 + self.pad), torch.full_like(ids, self.pad))
        h = self.model.decoder.roberta.embeddings(ids, pos, None)
        n = h.shape[0]

        def self_attention(li, q, k, v):
            self.kc[li][:, t] = k.view(n, H, 64)
            self.vc[li][:, t] = v.view(n, H, 64)
            return ops.attention(q, self.kc[li][:, :t + 1], self.vc[li][:, :t + 1], mask=None, causal=False)
        return self._layers(h, self_attention)

    def reorder(self, parents, t):
        """keep the caches of the surviving beams' parents (positions 0..t are filled)"""
        if self.graph is not None:
            self.g_parents.copy_(parents)          # applied at the head of the next replay
            return
        for li in range(len(self.layers)):
            self.kc[li][:, :t + 1] = self.kc[li][:, :t + 1].index_select(0, parents)
            self.vc[li][:, :t + 1] = self.vc[li][:, :t + 1].index_select(0, parents)

    # ---- captured step ---------------------------------------------------------------------------------------
    def _caches(self):
        return self.kvc if self.fast else self.kc + self.vc

    def _use_caches(self, tensors):
        if self.fast:
            self.kvc = list(tensors)
        else:
            self.kc, self.vc = list(tensors[:len(tensors) // 2]), list(tensors[len(tensors) // 2:])

    def _graph_step(self, previous):
        """previous: the cache set the step before wrote (re-ordered by the parents INTO the current set, which
        halves the traffic of re-ordering in place through a temporary), or None without beams"""
        H = self.H
        n = self.g_tok.shape[0]
        if previous is not None:
            for src, dst in zip(previous, self._caches()):
                torch.index_select(src, 0, self.g_parents, out=dst)
        if self.gather:      # beam i now continues the history of beam parents[i]; what it writes at position t lands in row i
            self.anc.copy_(self.anc.index_select(0, self.g_parents))
            self.anc.index_copy_(1, self.g_t, self.g_slot)
        self.g_mask.index_fill_(1, self.g_t, 0.0)
        ids = self.g_tok[:, None]
        pos = torch.where(ids.ne(self.pad), (self.g_t + (1 + self.pad)).expand_as(ids), torch.full_like(ids, self.pad))
        h = self.model.decoder.roberta.embeddings(ids, pos, None)
        if self.fast:
            return self._fast_layers(h)

        def self_attention(li, q, k, v):
            self.kc[li].index_copy_(1, self.g_t, k.view(n, 1, H, 64).to(self.kc[li].dtype))
            self.vc[li].index_copy_(1, self.g_t, v.view(n, 1, H, 64).to(self.vc[li].dtype))
            return ops.attention(q, self.kc[li], self.vc[li], mask=self.g_mask, causal=False)
        return self._layers(h, self_attention)

    def _prepare_fast(self, enc, n, max_length):
        lin, bf = torch.nn.functional.linear, torch.bfloat16
        B, L, _ = enc.shape
        H = self.H

        def w(*mods):      # bf16 weight and bias of one Linear, or of several stacked along the outputs
            return (torch.cat([m.weight.detach() for m in mods]).to(bf), torch.cat([m.bias.detach() for m in mods]).to(bf))
        self.w, self.kvx, self.kvc = [], [], []
        enc16 = enc.to(bf)
        for ly in self.layers:
            at, ca = ly.attention, ly.crossattention
            self.w.append({"q": w(at.self.query), "kv": w(at.self.key, at.self.value), "qkv": w(at.self.query, at.self.key, at.self.value),
                           "o": at.output.dense.weight.detach().to(bf),
                           "qx": w(ca.self.query), "ox": ca.output.dense.weight.detach().to(bf), "i": w(ly.intermediate.dense),
                           "out": ly.output.dense.weight.detach().to(bf)})
            self.kvx.append(lin(enc16, *w(ca.self.key, ca.self.value)).view(B, L, 2, H, 64))
            self.kvc.append(torch.zeros((n, max_length, 2, H, 64), dtype=bf, device=enc.device))
        head = self.model.decoder.lm_head
        # the vocabulary projection stays fp32: the log-probabilities rank the beams
        self.w_head = (w(head.dense), (head.decoder.weight.detach().float(), head.decoder.bias.detach().float()))

    def _fast_layers(self, h):
        """the decoder layers of `_layers` on the prepared bf16 weights: fp32 residual stream, bf16 copies of it for the
        GEMMs written by the LayerNorm kernel, dense biases added inside that kernel"""
        lin, H = torch.nn.functional.linear, self.H
        n = h.shape[0]
        h16 = h.to(torch.bfloat16)
        for li, ly in enumerate(self.layers):
            w, at, ca = self.w[li], ly.attention, ly.crossattention
            if self.gather:
                qkv = lin(h16, *w["qkv"]).view(n, 1, 3, H, 64)                     # one GEMM; q is read in place (strided)
                self.kvc[li].index_copy_(1, self.g_t, qkv[:, :, 1:])
                ctx = ops.attention_decode_gather(qkv[:, :, 0], self.kvc[li], self.anc, self.g_t)
            else:
                q = lin(h16, *w["q"]).view(n, 1, H, 64)
                self.kvc[li].index_copy_(1, self.g_t, lin(h16, *w["kv"]).view(n, 1, 2, H, 64))
                ctx = ops.attention_q_kv(q, self.kvc[li], mask=self.g_mask)
            h, h16 = ops.add_layernorm(lin(ctx, w["o"]), h, at.output.LayerNorm.weight, at.output.LayerNorm.bias, at.eps,
                                       dual=True, bias=at.output.dense.bias)
            q = lin(h16, *w["qx"]).view(n // self.expand, self.expand, H, 64)
            ctx = ops.attention_q_kv(q, self.kvx[li], mask=self.key).view(n, 1, H * 64)
            h, h16 = ops.add_layernorm(lin(ctx, w["ox"]), h, ca.output.LayerNorm.weight, ca.output.LayerNorm.bias, ca.eps,
                                       dual=True, bias=ca.output.dense.bias)
            f = torch.nn.functional.gelu(lin(h16, *w["i"]))
            h, h16 = ops.add_layernorm(lin(f, w["out"]), h, ly.output.LayerNorm.weight, ly.output.LayerNorm.bias, ly.eps,
                                       dual=True, bias=ly.output.dense.bias)
        head = self.model.decoder.lm_head
        x = torch.nn.functional.gelu(lin(h16, *self.w_head[0]))
        x = ops.add_layernorm(x, None, head.layer_norm.weight, head.layer_norm.bias, head.eps)
        return torch.log_softmax(lin(x[:, -1].float(), *self.w_head[1]), dim=-1)

    def _capture(self, n, max_length, dev, reorder):
        self.g_tok = torch.zeros(n, dtype=torch.long, device=dev)
        self.g_t = torch.zeros(1, dtype=torch.long, device=dev)
        self.g_parents = torch.arange(n, dtype=torch.long, device=dev)
        self.g_mask = torch.full((n, max_length), torch.finfo(torch.float32).min, dtype=torch.float32, device=dev)
        # inside a capture autocast must not cache its casts (they would belong to the graph's pool)
        if self.fast:
            ctx = lambda: torch.autocast("cuda", enabled=False)   # noqa: E731
        elif torch.is_autocast_enabled("cuda"):
            ctx = lambda: torch.autocast("cuda", dtype=torch.get_autocast_dtype("cuda"), cache_enabled=False)   # noqa: E731
        else:
            ctx = contextlib.nullcontext
        # ONE warm-up stream per device for the life of the process: the BLAS library keeps a workspace (76 MB here)
        # per stream it has ever run on, so a fresh stream per call would pin another one each time
        side = _warmup_streams.get(dev.index)
        if side is None:
            side = _warmup_streams[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        if self.gather:
            self.anc = torch.arange(n, dtype=torch.int32, device=dev)[:, None].repeat(1, max_length)     # anc[i][s]: see _graph_step
            self.g_slot = torch.arange(n, dtype=torch.int32, device=dev)[:, None]
            reorder = False
        sets = [self._caches()]
        if reorder:
            sets.append([torch.zeros_like(c) for c in sets[0]])
        with torch.cuda.stream(side), ctx():       # warm-up off the capture: library handles, kernel attributes
            for _ in range(2):
                self._graph_step(sets[1] if reorder else None)
        torch.cuda.current_stream(dev).wait_stream(side)
        graphs, self.g_logp = [], []
        for which in range(len(sets)):             # position t replays graph t % 2: reads set (t + 1) % 2, writes set t % 2
            self._use_caches(sets[which])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=graphs[0].pool() if graphs else None), ctx():
                self.g_logp.append(self._graph_step(sets[1 - which] if reorder else None))
            graphs.append(g)
        self.g_mask.fill_(torch.finfo(torch.float32).min)      # the warm-up opened column 0
        if self.gather:
            self.anc.copy_(torch.arange(n, dtype=torch.int32, device=dev)[:, None].expand(n, max_length))
        self.graph, self.g_sets = graphs, sets     # both cache sets stay alive as long as the graphs that write them


@torch.no_grad()
def generate(model, input_ids, attention_mask=None, num_beams=1, num_return_sequences=None, max_length=20,
             length_penalty=1.0, early_stopping=False, bos_token_id=None, eos_token_id=None, pad_token_id=0, graph=None):
    """-> (sequences [B * num_return_sequences, T] int64, sequences_scores [B * num_return_sequences] or None).
    Same argument meaning as the `generate` call at main.py:218-226; greedy search for num_beams == 1
    (sequences_scores is None then, as with Hugging Face, and main.py:228-231 reports zeros).
    graph: replay the decode step from a captured HIP graph (None = yes on a GPU with the HIP ops)."""
    was_training = model.training
    model.eval()
    try:
        if num_beams == 1:
            return _greedy(model, input_ids, attention_mask, max_length, bos_token_id, eos_token_id, pad_token_id, graph), None
        return _beam_search(model, input_ids, attention_mask, num_beams, num_return_sequences or 1, max_length,
                            length_penalty, early_stopping, bos_token_id, eos_token_id, pad_token_id, graph)
    finally:
        model.train(was_training)


def _greedy(model, input_ids, attention_mask, max_length, bos, eos, pad, graph=None):
    B, dev = input_ids.shape[0], input_ids.device
    st = _DecoderState(model, input_ids, attention_mask, 1, max_length, _use_graph(model, dev, graph))
    seqs = torch.full((B, 1), bos, dtype=torch.long, device=dev)
    unfinished = torch.ones(B, dtype=torch.long, device=dev)
    for t in range(max_length - 1):
        nxt = st.step(seqs[:, -1], t).argmax(dim=-1)
        nxt = nxt * unfinished + pad * (1 - unfinished)
        seqs = torch.cat([seqs, nxt[:, None]], dim=1)
        if eos is not None:
            unfinished = unfinished * nxt.ne(eos).long()
        if int(unfinished.max()) == 0:
            break
    return seqs


def _beam_search(model, input_ids, attention_mask, nb, keep, max_length, length_penalty, early_stopping, bos, eos, pad,
                 graph=None):
    B, dev = input_ids.shape[0], input_ids.device
    st = _DecoderState(model, input_ids, attention_mask, nb, max_length, _use_graph(model, dev, graph))
    seqs = torch.full((B * nb, 1), bos, dtype=torch.long, device=dev)
    beam_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9                        # all beams start identical: only the first one counts
    beam_scores = beam_scores.view(-1)
    hyps = [_BeamHypotheses(nb, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    for t in range(max_length - 1):
        cur_len = seqs.shape[1]
        logp = st.step(seqs[:, -1], t)
        vocab = logp.shape[-1]
        # the 2 * num_beams best continuations of each input, in two stages (per beam, then across an input's beams):
        # one top-k over [B, num_beams * vocab] runs on B workgroups only
        k1 = min(2 * nb, vocab)
        s1, i1 = torch.topk(logp + beam_scores[:, None], k1, dim=1, largest=True, sorted=True)
        top_s, j = torch.topk(s1.view(B, nb * k1), 2 * nb, dim=1, largest=True, sorted=True)
        top_beam = torch.div(j, k1, rounding_mode="floor")
        top_tok = i1.view(B, nb * k1).gather(1, j)
        # BeamSearchScorer.process, on the host like the original (B * 2 * nb scalars per step)
        s_l, b_l, t_l = top_s.tolist(), top_beam.tolist(), top_tok.tolist()
        seqs_host = None
        nscore, ntok, nidx = [0.0] * (B * nb), [pad] * (B * nb), [0] * (B * nb)     # finished items: zeros / pad / beam 0
        for b in range(B):
            if done[b]:
                continue
            slot, tb, sb, bb = b * nb, t_l[b], s_l[b], b_l[b]
            for rank in range(2 * nb):
                tok = tb[rank]
                if eos is not None and tok == eos:
                    if rank >= nb:                             # an end token outside the top num_beams is ignored
                        continue
                    if seqs_host is None:
                        seqs_host = seqs.cpu()
                    hyps[b].add(seqs_host[b * nb + bb[rank]].clone(), sb[rank])
                else:
                    nscore[slot], ntok[slot], nidx[slot] = sb[rank], tok, b * nb + bb[rank]
                    slot += 1
                if slot == (b + 1) * nb:
                    break
            if slot < (b + 1) * nb:
                raise ValueError("fewer than num_beams non-end candidates: increase the vocabulary or lower num_beams")
            done[b] = done[b] or hyps[b].is_done(max(sb), cur_len)
        beam_scores = torch.tensor(nscore, dtype=torch.float32).to(dev)
        tok_par = torch.tensor([ntok, nidx], dtype=torch.long).to(dev)
        parents = tok_par[1]
        st.reorder(parents, t)
        seqs = torch.cat([seqs.index_select(0, parents), tok_par[0][:, None]], dim=1)
        if all(done) or seqs.shape[1] >= max_length:
            break
    # BeamSearchScorer.finalize
    seqs_host, final = seqs.cpu(), beam_scores.cpu().tolist()
    for b in range(B):
        if done[b]:
            continue
        for j in range(nb):
            hyps[b].add(seqs_host[b * nb + j], final[b * nb + j])
    best, best_scores = [], []
    for b in range(B):
        ranked = sorted(hyps[b].beams, key=lambda x: x[0])
        for _ in range(keep):
            sc, hyp = ranked.pop()
            best.append(hyp); best_scores.append(sc)
    lengths = [int(h.shape[0]) for h in best]
    width = min(max(lengths) + 1, max_length)
    out = torch.full((len(best), width), pad, dtype=torch.long)
    for i, h in enumerate(best):
        out[i, :lengths[i]] = h
        if lengths[i] < width:
            out[i, lengths[i]] = eos
    return out.to(dev), torch.tensor(best_scores, dtype=torch.float32, device=dev)
