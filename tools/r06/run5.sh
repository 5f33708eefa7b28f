timeout 2400 python -m pytest tests/test_knn_gpu.py tests/test_host_path.py tests/test_knn_hostile_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error" | tail -8
