TRX_LIB=libtrxknn_lab.so timeout 900 python -m pytest tools/experiments/lab_checks_knn.py -q -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
TRX_NN_LIB=libtrxnn_lab.so timeout 900 python -m pytest tools/experiments/lab_checks.py -q -x 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
timeout 900 python -m pytest tests/test_knn_gpu.py -q -x -m gpu -k "crowd or bootstrap or c0" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
