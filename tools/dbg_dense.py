import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd import dense
from textreact_amd.predictor.model import Config
from textreact_amd.predictor import ops
cfg = Config(vocab_size=500, num_hidden_layers=2, max_position_embeddings=64)
torch.manual_seed(0)
enc = dense.DenseEncoder(cfg).cuda().eval()
g = torch.Generator().manual_seed(3)
c_ids = torch.randint(1, 500, (8, 40), generator=g); c_am = torch.ones_like(c_ids); c_am[::4, 25:] = 0
for ac in (True, False):
    for be in ("hip", "torch"):
        enc.backend = be
        e = dense.encode(enc, c_ids, c_am, autocast=ac, out_dtype=torch.float32)
        print("autocast", ac, be, "nan", int(torch.isnan(e).sum()), "absmax", float(e.abs().max()))
# op level
q = torch.randn(2, 40, 12, 64, device="cuda").bfloat16()
m = torch.zeros(2, 40, device="cuda"); m[0, 25:] = torch.finfo(torch.float32).min
o = ops.attention(q, q, q, mask=m)
print("attn nan", int(torch.isnan(o).sum()))
x = torch.randn(80, 768, device="cuda").bfloat16()
y = ops.add_layernorm(x, x, torch.ones(768, device="cuda"), torch.zeros(768, device="cuda"), 1e-12)
print("ln nan", int(torch.isnan(y).sum()))
