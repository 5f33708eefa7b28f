timeout 1200 python -m pytest tests/test_knn_hostile_gpu.py -q -m gpu -x --timeout 300 2>&1 | grep -v "^  File\|dist-packages" | head -60
