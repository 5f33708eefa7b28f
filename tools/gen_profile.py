"""where one beam-search step spends its time: host (cProfile, sorted by own time) + wall split replay / sync / scorer"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textreact_amd.predictor.generate import generate  # noqa: E402
from textreact_amd.predictor.model import Config, TextReactModel  # noqa: E402

B, L, nb, T = 8, 512, 20, 160
g = torch.Generator().manual_seed(0)
ids = torch.randint(1, 31090, (B, L), generator=g).cuda()
am = torch.ones(B, L, dtype=torch.long, device="cuda")
torch.manual_seed(0)
m = TextReactModel(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5,
                                                    is_decoder=True), backend="hip").cuda().eval()


def run():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return generate(m, ids, am, num_beams=nb, num_return_sequences=nb, max_length=T, length_penalty=0,
                        bos_token_id=12, eos_token_id=13, pad_token_id=0, graph=True)


run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); print("wall ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile()
pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
