"""Dense embedding producer + on-device retrieval: the step BEFORE the flat index.

The reference obtains its 768-d text / reaction embeddings from an external dense retriever
(README.md:44-47: a Tevatron bi-encoder -- BERT encoder, the [CLS] hidden state as the embedding,
dot-product similarity), writes them to disk, searches them with FAISS and converts the ranking with
retrieve/convert_format.py:6-16.  Here the same three steps stay on the GPU: the BERT encoder of
`predictor/model.py` (attention and LayerNorm through libtrxnn.so) produces [N, 768] embeddings
directly in HBM, they are added to the flat index (libtrxknn.so) as bf16 without a host round trip,
and the neighbours come back as the neighbor-file structure `textreact/dataset.py:40-44` reads.
Tokenisation stays outside (the reference's tokenizer files are not part of the repo): inputs are
token ids + attention masks.

    enc = DenseEncoder(Config(vocab_size=31090)).cuda().eval()      # BERT-base shape = SciBERT's
    enc.load_state_dict(tevatron_state_dict, strict=False)            # HF parameter names
    corpus = encode(enc, corpus_ids, corpus_mask)                      # [N, 768] bf16 on the GPU
    index = build_index(corpus)                                        # IndexFlatIP in HBM
    result = retrieve(enc, index, query_ids, query_mask, query_keys, corpus_keys, k=10)
    neighbors.write_neighbors("train.json", result)
"""
import torch
import torch.nn as nn

from . import faiss_compat as faiss
from .neighbors import build_result
from .predictor.model import BertEncoder, Config, additive_key_mask  # noqa: F401  (Config re-exported)


class DenseEncoder(nn.Module):
    """BERT encoder + [CLS] pooling (Tevatron's DenseModel.encode_passage / encode_query without a
    projection head).  Parameter names are the Hugging Face ones under `encoder.` exactly as in
    TextReactModel, so a predictor checkpoint's encoder or a Tevatron `model.lm_q` / `lm_p`
    state dict loads with the prefix renamed."""

    def __init__(self, cfg, normalize=False, encoder=None):
        super().__init__()
        self.encoder = encoder if encoder is not None else BertEncoder(cfg)
        self.normalize = normalize

    @classmethod
    def wrap(cls, encoder, normalize=False):
        """[CLS] pooling over an EXISTING encoder module (the predictor's own, textreact_amd.live): no parameter is copied,
        the embeddings follow the weights as they train"""
        return cls(None, normalize=normalize, encoder=encoder)

    def forward(self, input_ids, attention_mask=None):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        h = self.encoder(input_ids, additive_key_mask(attention_mask), None, None, None)
        cls = h[:, 0]
        return torch.nn.functional.normalize(cls.float(), dim=-1).to(cls.dtype) if self.normalize else cls


@torch.no_grad()
def encode(model, input_ids, attention_mask=None, batch_size=256, out_dtype=torch.bfloat16, autocast=True, lengths=None,
           batch_tokens=65536):
    """[N, L] token ids -> [N, hidden] embeddings on the model's device, in batches; eval mode, and by
    default bf16 autocast (the matrix-core attention path).  The ids may live on the host (up to 1 GiB of them are moved
    at once, more a batch at a time).

    lengths [N] (tokens per row, special tokens included; padding behind them): the rows are encoded longest first and
    every batch is cut to ITS longest row -- a corpus of passages of 16 ... 130 tokens padded to 130 is 44 % padding, and
    the encoder's cost is linear in the padded width (a key mask hides padding from the attention, it does not make it
    free) -- and holds as many rows as fit `batch_tokens` padded tokens (a multiple of 64, at least `batch_size`): the
    encoder's GEMMs see ~65,000 rows whatever the width (256 rows of 74 tokens are 19,000: 46.9 k passages/s of 16 ... 128
    tokens through BERT-base on one MI355X; 65,536 tokens a batch: 55.6 k; tools/encode_profile.py).  The embedding of a row does not depend on the width it was padded to, nor on its batch,
    beyond the rounding of the library GEMMs, whose tile choice may change with the batch shape."""
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()
    n = input_ids.shape[0]
    out = torch.empty((n, model.encoder.embeddings.word_embeddings.embedding_dim), dtype=out_dtype, device=dev)
    if input_ids.device != dev and input_ids.numel() * input_ids.element_size() <= (1 << 30):
        # (row gathers of a host tensor cost tens of milliseconds a batch where torch's CPU thread pool is larger than the
        # cores the process may use; one copy of the ids is a few hundred megabytes at most)
        input_ids = input_ids.to(dev)
        attention_mask = None if attention_mask is None else attention_mask.to(dev)
    order = None
    if lengths is not None and n:
        lengths = torch.as_tensor(lengths).to(input_ids.device)
        order = torch.argsort(lengths, descending=True, stable=True)
        sorted_len = lengths[order].clamp(min=1, max=input_ids.shape[1]).tolist()      # one host read
        bounds, lo = [], 0
        while lo < n:
            rows = max(batch_size, (batch_tokens // sorted_len[lo]) // 64 * 64) if batch_tokens else batch_size
            bounds.append((lo, min(n, lo + rows), sorted_len[lo]))
            lo += rows
    else:
        bounds = [(lo, min(n, lo + batch_size), None) for lo in range(0, n, batch_size)]
    import contextlib
    # ONE autocast block around all batches: its cache holds the bf16 casts of the weights, so they are made once per call
    # and not once per batch (~75 cast launches and half a gigabyte of traffic a batch for BERT-base)
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if (autocast and dev.type == "cuda") else contextlib.nullcontext()
    from .predictor import ops
    try:
        with ctx, ops.packed_projections():      # (the packed bf16 q / k / v weight of a layer: made once per call, like the casts)
            for lo, hi, width in bounds:
                if order is None:
                    ids = input_ids[lo:hi].to(dev)
                    am = None if attention_mask is None else attention_mask[lo:hi].to(dev)
                else:
                    rows = order[lo:hi]
                    ids = input_ids[rows, :width].to(dev)
                    am = None if attention_mask is None else attention_mask[rows, :width].to(dev)
                e = model(ids, am)
                if order is None:
                    out[lo:hi] = e.to(out_dtype)
                else:
                    out[rows.to(dev)] = e.to(out_dtype)
    finally:
        model.train(was_training)
    return out


def build_index(embeddings, metric=faiss.METRIC_INNER_PRODUCT):
    """flat index over embeddings that are already on the GPU (bf16 / fp32): no host copy"""
    index = faiss.IndexFlat(embeddings.shape[1], metric, device=embeddings.device.index or 0)
    index.add(embeddings)
    return index


def retrieve(model, index, query_ids, query_mask, query_keys, corpus_keys, k=10, batch_size=256):
    """encode the queries, search, and return the neighbor-file structure
    [{'id': query key, 'nn': [corpus key, ...]}] (retrieve_faiss.py:116; convert_format.py:6-16)."""
    q = encode(model, query_ids, query_mask, batch_size)
    _, rank = index.search(q, k)
    return build_result(query_keys, rank.cpu().numpy(), corpus_keys)
