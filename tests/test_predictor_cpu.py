"""Predictor module tree on CPU: parameter names and eval-mode logits against the golden produced
by the reference's own get_model (tests/golden/make_predictor_golden.py).  Uses the fp32 PyTorch
statement of the two ops (oracle/nn_ref.py, injected by the `reference_ops` fixture; the HIP kernels are checked
against the same statement in tests/test_predictor_gpu.py); the product itself must refuse to run without a GPU."""
import json
import os

import numpy as np
import pytest
import torch

from textreact_amd.predictor import ops
from textreact_amd.predictor.model import Config, TextReactModel, random_state_dict

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "predictor_small.npz")


def _load():
    z = np.load(G)
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
    m.load_state_dict(random_state_dict(m, int(z["seed"])))
    m.eval()
    return z, m


def test_state_dict_names_match_the_reference():
    z, m = _load()
    assert sorted(m.state_dict().keys()) == json.loads(str(z["state_dict_keys"]))
    # the names SURVEY.md 5.4 lists for the Lightning checkpoint (prefix `model.` added by the LightningModule)
    keys = set(m.state_dict().keys())
    for k in ("encoder.embeddings.word_embeddings.weight", "encoder.encoder.layer.1.attention.self.query.weight",
              "encoder.encoder.layer.0.attention.output.LayerNorm.bias", "encoder.pooler.dense.weight",
              "decoder.roberta.encoder.layer.0.crossattention.self.key.bias",
              "decoder.roberta.encoder.layer.1.output.LayerNorm.weight", "decoder.lm_head.layer_norm.weight",
              "decoder.lm_head.decoder.weight", "decoder.lm_head.bias"):
        assert k in keys, k


def test_logits_match_reference_golden_fp32(reference_ops):
    z, m = _load()
    t = lambda k: torch.from_numpy(z[k])
    with torch.no_grad():
        logits, enc = m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))
    assert float((logits - t("logits")).abs().max()) <= 1e-3          # north-star tolerance
    assert float((enc - t("encoder_last_hidden_state")).abs().max()) <= 1e-3


def test_full_size_logits_match_the_reference_golden(reference_ops):
    """BERT-base / bert_l6.json, L = 512, T = 7: the module tree (on the PyTorch statement of the ops) against logits of the
    reference's own get_model at the scripts' size; the fixture also carries the SHA-256 of the full arrays"""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_predictor_golden import full_inputs
    z = np.load(os.path.join(os.path.dirname(G), "predictor_full.npz"))
    enc, dec = json.loads(str(z["enc_cfg"])), json.loads(str(z["dec_cfg"]))
    m = TextReactModel(Config(**enc), Config(is_decoder=True, **dec))
    m.load_state_dict(random_state_dict(m, int(z["seed"])))
    ids, am, dids = full_inputs(7)
    with torch.no_grad():
        logits, states = m.eval()(ids, am, dids)
    assert float((logits - torch.from_numpy(z["logits_T7"])).abs().max()) <= 1e-5
    pos = z["enc_pos"]
    assert np.abs(states.numpy()[pos[:, 0], pos[:, 1]] - z["enc_at"]).max() <= 1e-5
    if hashlib.sha256(logits.numpy().tobytes()).hexdigest() != str(z["sha_logits_T7"]):
        import warnings
        warnings.warn("full-size logits agree to 1e-5 but not bit for bit with the fixture's hash (BLAS threading?)")


def test_the_product_refuses_cpu_tensors():
    z, m = _load()
    t = lambda k: torch.from_numpy(z[k])
    with pytest.raises(ops.TrxNNError):
        m(t("input_ids"), t("attention_mask"), t("decoder_input_ids"), t("decoder_attention_mask"))


def test_nn_library_exports_the_header_symbols():
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "trx_nn.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(trx_[a-z0-9_]+)\s*\(", src)))
    assert declared == sorted(ops.SYMBOLS)
    L = ops.lib()
    for s in declared:
        assert hasattr(L, s)
    assert L.trx_attention_fwd(None, None, None, None, 0, 0, 1, 1, 1, 1, 1.0, 0, None, None) == -1
    assert b"bad argument" in L.trx_nn_last_error()


def test_embedding_tables_grow_and_keep_their_rows(reference_ops):
    import torch
    from textreact_amd.predictor.model import (Config, TextReactModel, expand_position_embeddings, expand_word_embeddings,
                                               gather_prediction_each_neighbor)
    m = TextReactModel(Config(vocab_size=50, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                              max_position_embeddings=16),
                       Config(vocab_size=20, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                              max_position_embeddings=16, is_decoder=True)).eval()
    ids = torch.randint(1, 50, (2, 16)); dids = torch.randint(1, 20, (2, 5))
    with torch.no_grad():
        a = m(ids, None, dids)[0]
    pos, word = m.encoder.embeddings.position_embeddings.weight.clone(), m.encoder.embeddings.word_embeddings.weight.clone()
    expand_position_embeddings(m.encoder, 40); expand_word_embeddings(m.encoder, 70)
    expand_position_embeddings(m.encoder, 8)                              # never shrinks
    assert m.encoder.embeddings.position_embeddings.weight.shape[0] == 40 and m.encoder.embeddings.word_embeddings.weight.shape[0] == 70
    assert torch.equal(m.encoder.embeddings.position_embeddings.weight[:16], pos)
    assert torch.equal(m.encoder.embeddings.word_embeddings.weight[:50], word)
    with torch.no_grad():
        assert torch.equal(a, m(ids, None, dids)[0])                       # old inputs, same outputs
        m(torch.randint(50, 70, (2, 40)), None, dids)                      # the new rows are usable
    merged = gather_prediction_each_neighbor({0: {"prediction": ["a"], "score": [1.0]}, 1: {"prediction": ["b"], "score": [.5]},
                                              2: {"prediction": ["c"], "score": [.2]}}, 2)
    assert merged == {0: {"prediction": ["a", "b"], "score": [1.0, .5]}, 1: {"prediction": ["c"], "score": [.2]}}


def test_top_k_accuracies_follow_evaluate_py():
    import pandas as pd
    from textreact_amd.predictor.evaluate import CONDITION_COLS, evaluate_reaction_condition, evaluate_retrosynthesis
    df = pd.DataFrame([["c", "s1", "", "r1", ""], ["", "s2", "s3", "r2", "r3"], ["", "", "", "", ""]], columns=CONDITION_COLS)
    pred = {0: {"prediction": [["x", "", "", "", ""], ["c", "s1", "", "r1", ""]]},      # hit at rank 2
            1: {"prediction": [["", "s2", "s3", "r2", "r3"]]}}                           # hit at rank 1; example 2 unpredicted
    acc = evaluate_reaction_condition(pred, df)
    assert acc == {1: 1 / 3, 3: 2 / 3, 5: 2 / 3, 10: 2 / 3, 15: 2 / 3}
    retro = evaluate_retrosynthesis({0: {"prediction": ["CCO", "CC"]}, 1: {"prediction": ["N", "O", "C"]}}, ["CC", "S"],
                                    canonical=lambda s: s)
    assert retro == {1: 0.0, 2: 0.5, 3: 0.5, 5: 0.5, 10: 0.5, 20: 0.5}
