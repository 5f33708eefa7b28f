"""does PyTorch's TunableOp (hipBLASLt / rocBLAS solution search per GEMM shape) move the predictor's train step?
    python tools/tunable_probe.py [T]      prints ms per step: default library choice, then with torch.cuda.tunable on"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predictor as bp
from textreact_amd.predictor.model import Config
from textreact_amd.predictor import train

T = int(sys.argv[1]) if len(sys.argv) > 1 else 160
dev, B, L = torch.device("cuda", 0), 32, 512
g = torch.Generator().manual_seed(0)
batch = {"input_ids": torch.randint(1, 31090, (B, L), generator=g).to(dev), "attention_mask": torch.ones(B, L, dtype=torch.long, device=dev),
         "decoder_input_ids": torch.randint(14, 600, (B, T), generator=g).to(dev), "decoder_attention_mask": torch.ones(B, T, dtype=torch.long, device=dev)}
batch["attention_mask"][::3, L * 4 // 5:] = 0
torch.manual_seed(0)
p = train.Predictor(Config(vocab_size=31090), Config(vocab_size=600, num_hidden_layers=6, type_vocab_size=1, layer_norm_eps=1e-5, is_decoder=True), mlm=False).to(dev).train()
opt, _ = train.configure_optimizer(p, 1e-4, 0.01, 1000, 0.02)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _ = p.training_step(batch)
    loss.backward(); opt.step(); opt.zero_grad(set_to_none=True); train.mark_parameters_updated(p)
print("default  : %.2f ms" % bp.timeit(step, iters=10, warm=4), flush=True)
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True)
try:
    tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(20)
except Exception as e:
    print("limits:", e)
t0 = time.time()
for _ in range(3): step()
torch.cuda.synchronize()
print("tuning took %.1f s" % (time.time() - t0), flush=True)
tn.tuning_enable(False)
print("tunable  : %.2f ms" % bp.timeit(step, iters=10, warm=2), flush=True)
try:
    res = tn.get_results()
    print(len(res), "tuned entries; e.g.", res[:3])
except Exception as e:
    print("results:", e)
