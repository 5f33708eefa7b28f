"""time the train step only, hip and torch backends: python3 tools/train_time.py [bf16|fp16] [L T]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predictor as bp
dtype = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
L, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 160)
for r in bp.train_step_bench("cuda", steps=10, dtype=dtype, L=L, T=T):
    print(r["kernel"], r["backend"], r["dtype"], "L", L, "T", T, round(r["ms"], 2), "ms")
