"""generate(): the decode step replayed from a HIP graph vs eager launches vs PyTorch ops (bench_predictor.generate_bench alone)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_predictor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8          # --test_batch_size 32 in the RetroSyn scripts
for r in bench_predictor.generate_bench("cuda", B):
    print(json.dumps(r))
