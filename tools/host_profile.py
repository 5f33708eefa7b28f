"""where the host spends a training step (B = 4: the GPU is never the bottleneck)"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor.model import Config
from textreact_amd.predictor import train
dev = "cuda"
p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True),
                    mlm=False, backend="hip").to(dev).train()
opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
B, L, T = 4, 512, 160
g = torch.Generator().manual_seed(0)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = p.training_step(batch)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(4): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
