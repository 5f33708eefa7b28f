"""The text-augmented encoder-decoder of TextReact with the reference's parameter names.

Reference: textreact/model.py:21-31 builds `EncoderDecoderModel(BertModel, RobertaForCausalLM)`;
the decoder shape is textreact/configs/bert_l6.json.  This module tree owns its layers (the pinned
transformers 4.27.3 inlines attention in BertSelfAttention.forward, there is no hook to replace),
keeps every parameter name of the Hugging Face state dict -- `encoder.embeddings.*`,
`encoder.encoder.layer.N.attention.self.{query,key,value}`, `...attention.output.{dense,LayerNorm}`,
`...intermediate.dense`, `...output.{dense,LayerNorm}`, `decoder.roberta.*` with `crossattention`,
`decoder.lm_head.{bias,dense,layer_norm,decoder}` -- so a Lightning
checkpoint's `state_dict` (prefix `model.`) loads unchanged, and routes the two hot spots through
textreact_amd.predictor.ops: attention (self / cross / causal) and LayerNorm(dense(h) + residual).
Linear layers stay rocBLAS GEMMs (torch.nn.Linear); GELU is the exact erf form, as in BERT.

Training mode applies the reference's dropout (BERT defaults 0.1 / 0.1): on the attention
probabilities and on dense outputs before the residual LayerNorm -- both inside the fused kernels --
and on the embeddings after their LayerNorm (torch's dropout); eval mode is the identity, and parity
with the reference's logits is defined in eval mode.
"""
import os

import torch
import torch.nn as nn

from . import ops


class Config:
    """subset of the HF config fields that shape the network"""

    def __init__(self, vocab_size, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 pad_token_id=0, is_decoder=False, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **_):
        self.vocab_size, self.hidden_size, self.num_hidden_layers = vocab_size, hidden_size, num_hidden_layers
        self.num_attention_heads, self.intermediate_size = num_attention_heads, intermediate_size
        self.max_position_embeddings, self.type_vocab_size = max_position_embeddings, type_vocab_size
        self.layer_norm_eps, self.pad_token_id, self.is_decoder = layer_norm_eps, pad_token_id, is_decoder
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        assert hidden_size == num_attention_heads * 64, "the attention kernel is specialised for heads of 64"


class LayerNormParams(nn.Module):
    """holds `weight` / `bias` under the HF name; the math happens fused with the residual add"""

    def __init__(self, n):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))


class SelfAttentionProj(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.query = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.key = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.value = nn.Linear(cfg.hidden_size, cfg.hidden_size)


class AttnOutput(nn.Module):
    def __init__(self, cfg, in_features=None):
        super().__init__()
        self.dense = nn.Linear(in_features or cfg.hidden_size, cfg.hidden_size)
        self.LayerNorm = LayerNormParams(cfg.hidden_size)


def _dense_add_layernorm(dense, x, h, ln, eps, p):
    """(stream, low) = LayerNorm(dropout(dense(x)) + h).  Under bf16 autocast (x bf16, stream fp32)
    the dense layer runs WITHOUT its bias and the fused kernel adds it: the bias gradient then falls out of the
    LayerNorm backward instead of a separate reduction over all rows."""
    if x.dtype == torch.bfloat16 and h.dtype == torch.float32 and x.is_cuda:
        return ops.add_layernorm(ops.linear(x, dense.weight, None), h, ln.weight, ln.bias, eps,
                                 dropout_p=p, dual=True, bias=dense.bias)
    return ops.add_layernorm(dense(x), h, ln.weight, ln.bias, eps, dropout_p=p, dual=True)


class Attention(nn.Module):
    """`attention` / `crossattention` of a layer: self.{query,key,value} + output.{dense,LayerNorm}"""

    def __init__(self, cfg):
        super().__init__()
        self.self = SelfAttentionProj(cfg)
        self.output = AttnOutput(cfg)
        self.heads, self.eps = cfg.num_attention_heads, cfg.layer_norm_eps
        self.p_attn, self.p_hidden = cfg.attention_probs_dropout_prob, cfg.hidden_dropout_prob

    def forward(self, h, h_low, kv_low, mask, causal):
        """h: the residual stream; h_low / kv_low: what the projections read (a bf16 copy of the stream under
        autocast, the stream itself otherwise).  Returns the new (stream, low) pair."""
        B, Lq, Hd = h.shape
        Lk = kv_low.shape[1]
        pa = self.p_attn if self.training else 0.0
        sa = self.self
        if kv_low is h_low:
            # self-attention: ONE [*, 768] x [768, 2304] GEMM for q, k, v (same parameters, concatenated per call);
            # the attention kernels read the three slices of its output in place
            qkv = ops.linear_multi(h_low, (sa.query.weight, sa.key.weight, sa.value.weight),
                                   (sa.query.bias, sa.key.bias, sa.value.bias)).view(B, Lq, 3, self.heads, 64)
            ctx = ops.attention_qkv(qkv, mask=mask, causal=causal, dropout_p=pa)
        else:
            q = ops.linear(h_low, sa.query.weight, sa.query.bias).view(B, Lq, self.heads, 64)
            kv = ops.linear_multi(kv_low, (sa.key.weight, sa.value.weight), (sa.key.bias, sa.value.bias)).view(B, Lk, 2, self.heads, 64)
            ctx = ops.attention_q_kv(q, kv, mask=mask, causal=causal, dropout_p=pa)
        return _dense_add_layernorm(self.output.dense, ctx, h, self.output.LayerNorm, self.eps,
                                    self.p_hidden if self.training else 0.0)


class Intermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.intermediate_size)


class Layer(nn.Module):
    def __init__(self, cfg, cross=False):
        super().__init__()
        self.attention = Attention(cfg)
        if cross:
            self.crossattention = Attention(cfg)
        self.intermediate = Intermediate(cfg)
        self.output = AttnOutput(cfg, in_features=cfg.intermediate_size)
        self.eps, self.cross, self.p_hidden = cfg.layer_norm_eps, cross, cfg.hidden_dropout_prob

    def forward(self, h, h_low, self_mask, causal, enc_low, enc_mask):
        h, h_low = self.attention(h, h_low, h_low, self_mask, causal)
        if self.cross:
            h, h_low = self.crossattention(h, h_low, enc_low, enc_mask, False)
        f = torch.nn.functional.gelu(ops.linear(h_low, self.intermediate.dense.weight, self.intermediate.dense.bias))
        return _dense_add_layernorm(self.output.dense, f, h, self.output.LayerNorm, self.eps,
                                    self.p_hidden if self.training else 0.0)


class LayerStack(nn.Module):
    def __init__(self, cfg, cross=False):
        super().__init__()
        self.layer = nn.ModuleList([Layer(cfg, cross) for _ in range(cfg.num_hidden_layers)])


class Embeddings(nn.Module):
    def __init__(self, cfg, roberta=False):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg.vocab_size, cfg.hidden_size, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size,
                                                padding_idx=cfg.pad_token_id if roberta else None)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, cfg.hidden_size)
        self.LayerNorm = LayerNormParams(cfg.hidden_size)
        self.eps, self.roberta, self.pad = cfg.layer_norm_eps, roberta, cfg.pad_token_id
        self.p_hidden = cfg.hidden_dropout_prob

    def forward(self, input_ids, position_ids, token_type_ids):
        # The defaults of the reference's calls (no token types: all zero; BERT positions: 0 .. L-1 for every sample) are rows
        # broadcast over the batch, not gathers: the same sums in the same order ((word + type) + position, BertEmbeddings), but
        # the gradients of those two tables become plain reductions over the batch instead of two more sort-based
        # embedding backward passes (~0.13 ms each at 16,384 tokens: profiles/r03_train_step_kernel_stats.csv).
        if token_type_ids is None:
            e = self.word_embeddings(input_ids) + self.token_type_embeddings.weight[0]
        else:
            e = self.word_embeddings(input_ids) + self.token_type_embeddings(token_type_ids)
        if position_ids is None and not self.roberta:
            y = ops.add_layernorm(e + self.position_embeddings.weight[:input_ids.shape[1]], None, self.LayerNorm.weight,
                                  self.LayerNorm.bias, self.eps)
            return torch.nn.functional.dropout(y, self.p_hidden, self.training)
        if position_ids is None:       # RoBERTa: positions count non-pad tokens, starting at pad + 1
            nz = input_ids.ne(self.pad).int()
            position_ids = (torch.cumsum(nz, dim=1) * nz).long() + self.pad
        y = ops.add_layernorm(e, self.position_embeddings(position_ids), self.LayerNorm.weight,
                              self.LayerNorm.bias, self.eps)
        return torch.nn.functional.dropout(y, self.p_hidden, self.training)   # after the LayerNorm (BertEmbeddings)


class Pooler(nn.Module):
    """present in the checkpoint (encoder.pooler.dense.*); unused by the predictor's forward"""

    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)


class BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = Embeddings(cfg)
        self.encoder = LayerStack(cfg)
        self.pooler = Pooler(cfg)

    def forward(self, input_ids, key_mask, position_ids, token_type_ids, full_mask):
        h = self.embeddings(input_ids, position_ids, token_type_ids)
        m = full_mask if full_mask is not None else key_mask
        h_low = h
        for layer in self.encoder.layer:
            h, h_low = layer(h, h_low, m, False, None, None)
        self.last_low = h_low      # bf16 copy of the output under autocast (cross-attention reads it)
        return h


class RobertaBody(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = Embeddings(cfg, roberta=True)
        self.encoder = LayerStack(cfg, cross=True)


class LMHead(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = nn.Linear(cfg.hidden_size, cfg.hidden_size)
        self.layer_norm = LayerNormParams(cfg.hidden_size)
        self.decoder = nn.Linear(cfg.hidden_size, cfg.vocab_size)
        self.bias = nn.Parameter(torch.zeros(cfg.vocab_size))
        self.decoder.bias = self.bias      # tied, as in RobertaLMHead
        self.eps = cfg.layer_norm_eps

    def forward(self, h):
        # (ops.linear: with bf16 activations the two weight gradients join the backward pass's grouped launch -- the 600-row
        # vocabulary projection's cost the library 150 us as a product of its own)
        x = torch.nn.functional.gelu(ops.linear(h, self.dense.weight, self.dense.bias))
        x = ops.add_layernorm(x, None, self.layer_norm.weight, self.layer_norm.bias, self.eps)
        return ops.linear(x, self.decoder.weight, self.decoder.bias)


class RobertaCausalLM(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.roberta = RobertaBody(cfg)
        self.lm_head = LMHead(cfg)
        self.lm_head.decoder.weight = self.roberta.embeddings.word_embeddings.weight   # tie_word_embeddings


def additive_key_mask(attention_mask, dtype=torch.float32):
    """HF convention: (1 - mask) * finfo.min, one value per key"""
    return (1.0 - attention_mask.to(dtype)) * torch.finfo(dtype).min


def weight_shadows(model, like):
    """the model's ops.WeightShadows registry (created on first use) when a training forward under autocast on the GPU can
    use it, else None"""
    if not (like.is_cuda and torch.is_autocast_enabled("cuda") and torch.is_grad_enabled()) \
            or "TRX_NN_NO_SHADOWS" in os.environ:      # (the knob is for A/B timing)
        return None
    reg = model.__dict__.get("_weight_shadows")
    if reg is None:
        reg = model.__dict__["_weight_shadows"] = ops.WeightShadows()
    return reg


class TextReactModel(nn.Module):
    """forward(input_ids, attention_mask, decoder_input_ids, ...) -> (logits, encoder_last_hidden_state)"""

    def __init__(self, enc_cfg, dec_cfg):
        super().__init__()
        self.encoder = BertEncoder(enc_cfg)
        self.decoder = RobertaCausalLM(dec_cfg)

    def forward(self, input_ids, attention_mask=None, decoder_input_ids=None, decoder_attention_mask=None,
                position_ids=None, token_type_ids=None):
        if self.training:   # one generator draw for all the dropout sites of this pass, one multi-tensor cast of the Linear weights
            with ops.seed_scope(), ops.use_shadows(weight_shadows(self, input_ids)):
                return self._forward(input_ids, attention_mask, decoder_input_ids, decoder_attention_mask, position_ids, token_type_ids)
        return self._forward(input_ids, attention_mask, decoder_input_ids, decoder_attention_mask, position_ids, token_type_ids)

    def _forward(self, input_ids, attention_mask, decoder_input_ids, decoder_attention_mask, position_ids, token_type_ids):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        full = None
        if attention_mask.dim() == 3:      # --unattend_nonbonds: [B, L, L] mask (dataset.py:247-254)
            full = additive_key_mask(attention_mask)
            key = additive_key_mask(attention_mask.amax(dim=1))
        else:
            key = additive_key_mask(attention_mask)
        enc = self.encoder(input_ids, key, position_ids, token_type_ids, full)
        dmask = additive_key_mask(decoder_attention_mask) if decoder_attention_mask is not None else None
        enc_low = self.encoder.last_low
        self.encoder.last_low = None       # (a tensor kept on the module would keep this pass's autograd graph alive into the next one)
        h = self.decoder.roberta.embeddings(decoder_input_ids, None, None)
        h_low = h
        for layer in self.decoder.roberta.encoder.layer:
            h, h_low = layer(h, h_low, dmask, True, enc_low, key)
        return self.decoder.lm_head(h_low), enc


def random_state_dict(model, seed):
    """Deterministic fp32 weights for parity tests: every tensor drawn from a seeded CPU generator in
    state-dict order (N(0, 0.05); LayerNorm weights around 1).  Identical on every machine with this
    torch build, so a fixture needs to carry only the seed, not the weights."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = {}
    for name, t in model.state_dict().items():
        w = torch.randn(t.shape, generator=g, dtype=torch.float32) * 0.05
        if name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight"):
            w = w + 1.0
        sd[name] = w
    # tied tensors share one value
    sd["decoder.lm_head.decoder.weight"] = sd["decoder.roberta.embeddings.word_embeddings.weight"]
    sd["decoder.lm_head.decoder.bias"] = sd["decoder.lm_head.bias"]
    return sd


# ---- embedding-table growth (textreact/utils.py:18-44, called from model.py:32-35) -------------------------
def expand_position_embeddings(encoder, max_length):
    """grow the encoder's position table to max_length rows, old rows kept (new rows: nn.Embedding's init)"""
    emb = encoder.embeddings
    old = emb.position_embeddings.weight.data
    if max_length <= old.shape[0]:
        return
    new = nn.Embedding(max_length, old.shape[1]).to(old.device, old.dtype)
    new.weight.data[:old.shape[0]] = old
    emb.position_embeddings = new


def expand_word_embeddings(encoder, vocab_size):
    """grow the encoder's word table to vocab_size rows (--encoder_tokenizer smiles_text), old rows kept"""
    emb = encoder.embeddings
    old = emb.word_embeddings.weight.data
    if vocab_size <= old.shape[0]:
        return
    new = nn.Embedding(vocab_size, old.shape[1], padding_idx=emb.word_embeddings.padding_idx).to(old.device, old.dtype)
    new.weight.data[:old.shape[0]] = old
    emb.word_embeddings = new


def gather_prediction_each_neighbor(prediction, num_neighbors):
    """--test_each_neighbor: predictions of the num_neighbors copies of one example merged (utils.py:55-64)"""
    results = {}
    for i, pred in sorted(prediction.items()):
        idx = i // num_neighbors
        if idx not in results:
            results[idx] = pred
        else:
            for key in results[idx]:
                results[idx][key] += pred[key]
    return results
