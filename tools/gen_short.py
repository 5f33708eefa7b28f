"""short decodes (the RCR task: ~8 positions, --test_batch_size 64, --num_beams 15): is capturing a graph worth it?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textreact_amd.predictor.generate import generate  # noqa: E402
from textreact_amd.predictor.model import Config, TextReactModel  # noqa: E402

B, L, nb = 64, 512, 15
g = torch.Generator().manual_seed(0)
ids = torch.randint(1, 31090, (B, L), generator=g).cuda()
torch.manual_seed(0)
m = TextReactModel(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5,
                                                    is_decoder=True), backend="hip").cuda().eval()
for T in (4, 8, 16, 32, 128):
    for graph in (True, False):
        def run():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return generate(m, ids, None, num_beams=nb, num_return_sequences=nb, max_length=T, length_penalty=0, bos_token_id=12,
                                eos_token_id=13, pad_token_id=0, graph=graph)
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        print("max_length", T, "graph" if graph else "eager", round((time.perf_counter() - t0) / 3 * 1e3, 1), "ms", flush=True)
