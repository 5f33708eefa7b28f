set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_host_path.py -x -q -m gpu 2>&1 | tail -15
timeout 900 python -m pytest tests/test_knn_gpu.py -x -q -m gpu -k "int_inputs or golden or c0 or int8 or bit_vectors or chunk" 2>&1 | tail -5
for t in 8 16 32 64 128; do TRX_HOST_THREADS=$t timeout 300 python tools/host_path_probe.py 200000 2048 int64 | tee -a gpurun_out/r06/host_path_sweep.jsonl; done
timeout 300 python tools/host_path_probe.py 200000 2048 int64 | tee gpurun_out/r06/host_path.jsonl
timeout 300 python tools/host_path_probe.py 800000 1024 int8 | tee -a gpurun_out/r06/host_path.jsonl
timeout 300 python tools/host_path_probe.py 300000 768 float32 | tee -a gpurun_out/r06/host_path.jsonl
lscpu | head -25 > gpurun_out/r06/lscpu.txt; numactl -H >> gpurun_out/r06/lscpu.txt 2>&1
