"""a few optimisation steps of the full-size predictor on the hip backend: target for rocprofv3
(python3 tools/train_step_one.py [steps] [backend] [L T])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor.model import Config
from textreact_amd.predictor import train
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
backend = sys.argv[2] if len(sys.argv) > 2 else "hip"
dev = "cuda"
B = 32
L, T = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 160)
g = torch.Generator().manual_seed(0)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev),
         "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev),
         "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
batch["attention_mask"][::3, L * 4 // 5:] = 0
enc = Config(vocab_size=31090)
dec = Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True)
torch.manual_seed(0)
p = train.Predictor(enc, dec, mlm=False, backend=backend).to(dev).train()
opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
for _ in range(steps):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = p.training_step(batch)
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("loss", float(loss))
