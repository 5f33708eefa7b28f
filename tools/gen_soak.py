"""repeated generate() calls: graphs and their pools must be released with each call (memory stays flat)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textreact_amd.predictor.generate import generate  # noqa: E402
from textreact_amd.predictor.model import Config, TextReactModel  # noqa: E402

B, L, nb, T = 8, 512, 20, 40
g = torch.Generator().manual_seed(0)
torch.manual_seed(0)
m = TextReactModel(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5,
                                                    is_decoder=True), backend="hip").cuda().eval()
first = None
for it in range(40):
    ids = torch.randint(1, 31090, (B, L - (it % 5) * 17), generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=it % 2 == 0):
        seq, sc = generate(m, ids, None, num_beams=nb if it % 3 else 1, num_return_sequences=nb if it % 3 else 1, max_length=T,
                           length_penalty=0, bos_token_id=12, eos_token_id=13, pad_token_id=0)
    torch.cuda.synchronize()
    if it % 5 == 4:
        print(it, "allocated MB %.0f reserved MB %.0f" % (torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20), flush=True)
