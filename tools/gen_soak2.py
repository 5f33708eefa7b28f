import gc
import os
import sys
import weakref

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textreact_amd.predictor import generate as G  # noqa: E402
from textreact_amd.predictor.model import Config, TextReactModel  # noqa: E402

refs = []
orig = G._DecoderState.__init__


def patched(self, *a, **k):
    orig(self, *a, **k)
    refs.append(weakref.ref(self))


G._DecoderState.__init__ = patched
gc.disable()
torch.manual_seed(0)
m = TextReactModel(Config(vocab_size=31090, num_hidden_layers=2), Config(vocab_size=600, num_hidden_layers=2, type_vocab_size=1, layer_norm_eps=1e-5,
                                                                         is_decoder=True), backend="hip").cuda().eval()
g = torch.Generator().manual_seed(0)
for it in range(6):
    ids = torch.randint(1, 31090, (8, 128), generator=g).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=it % 2 == 0):
        G.generate(m, ids, None, num_beams=5, num_return_sequences=5, max_length=40, length_penalty=0, bos_token_id=12, eos_token_id=13,
                   pad_token_id=0, graph=it < 4)
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_allocated() / 2**20
    alive = refs[-1]() is not None
    if alive:
        print("referrers:", [type(r).__name__ for r in gc.get_referrers(refs[-1]())][:8])
    n = gc.collect()
    print(it, "graph" if it < 4 else "eager", "state alive without gc:", alive, "allocated MB %.0f -> %.0f after gc (%d objects)" % (a0, torch.cuda.memory_allocated() / 2**20, n), flush=True)
