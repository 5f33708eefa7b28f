"""time the hip-backend train step only (python3 tools/train_time.py)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predictor as bp
for r in bp.train_step_bench("cuda", steps=10):
    print(r["backend"], round(r["ms"], 2), "ms")
