"""fast (bf16 weights, graph) vs eager decode step: log-probabilities position by position on the same tokens"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textreact_amd.predictor.generate import _DecoderState  # noqa: E402
from textreact_amd.predictor.model import Config, TextReactModel, random_state_dict  # noqa: E402

torch.manual_seed(3)
m = TextReactModel(Config(vocab_size=300, num_hidden_layers=2), Config(vocab_size=40, num_hidden_layers=2, type_vocab_size=1,
                   layer_norm_eps=1e-5, is_decoder=True), backend="hip")
m.load_state_dict(random_state_dict(m, 5))          # BERT-style init (std 0.02): logits of a trained model's size
m = m.cuda().eval()
g = torch.Generator().manual_seed(1)
ids = torch.randint(1, 300, (3, 70), generator=g).cuda()
am = torch.ones(3, 70, dtype=torch.long).cuda()
am[1, 50:] = 0
nb, T = 5, 12
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    a = _DecoderState(m, ids, am, nb, T, graph=False)
    b = _DecoderState(m, ids, am, nb, T, graph=True)
    with torch.autocast("cuda", enabled=False):
        c = _DecoderState(m, ids, am, nb, T, graph=False)       # fp32 truth
    for t in range(T - 1):
        tok = torch.randint(3, 40, (3 * nb,), generator=g).cuda()
        la, lb = a.step(tok, t).clone(), b.step(tok, t).clone()
        with torch.autocast("cuda", enabled=False):
            lc = c.step(tok, t).clone()
        par = torch.arange(3 * nb).view(3, nb).flip(1).reshape(-1).cuda()
        a.reorder(par, t); b.reorder(par, t); c.reorder(par, t)
        print(t, "eager-fast %.4f  eager-fp32 %.4f  fast-fp32 %.4f" % (float((la - lb).abs().max()), float((la - lc).abs().max()),
                                                                      float((lb - lc).abs().max())))
