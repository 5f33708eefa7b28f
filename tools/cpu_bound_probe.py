"""is the training step limited by the host? wall time per step vs the time the host needs to ENQUEUE a step
(measured by not synchronising until many steps were issued)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from textreact_amd.predictor.model import Config
from textreact_amd.predictor import train
dev = "cuda"
B, L, T = 32, 512, 160
g = torch.Generator().manual_seed(0)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True),
                    mlm=False, backend=sys.argv[1] if len(sys.argv) > 1 else "hip").to(dev).train()
opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = p.training_step(batch)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()          # host finished enqueueing (if the GPU is the bottleneck the queue just grows)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step   wall %.2f ms/step   (host-bound if the two are equal)" % ((t1 - t0) * 100, (t2 - t0) * 100))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
