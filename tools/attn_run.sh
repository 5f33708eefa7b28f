timeout 300 python -m pytest tests/test_predictor_gpu.py -x -q -m gpu 2>&1 | tail -3; python bench_predictor.py 2>&1 | grep attention | grep bfloat | python3 -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['kernel'], j['what'], round(j['ms'],4), 'ms', round(j['roofline']['achieved'],1), 'TF/s', 'torch', round(j['torch_eager_fp32_ms'],3))
"
