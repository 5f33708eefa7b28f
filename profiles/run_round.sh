#!/bin/bash
# One gpurun call that takes every measurement a round commits under profiles/ (run it through gpurun; summaries are
# made afterwards, here, by profiles/summarize*.py from what gpurun merged back into gpurun_out/).
#   profiles/run_round.sh <tag>
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/gputest.log 2>&1; echo "rc=$?" >> $O/gputest.log
python bench.py > $O/bench.jsonl 2> $O/bench.err
python bench.py --steps 20 --warmup 2 --no-cpu-baseline >> $O/bench.jsonl 2>> $O/bench.err
python bench.py --workload fingerprint --host-api > $O/fingerprint_bench.jsonl 2> $O/fingerprint_bench.err
python bench.py --workload morgan --n-corpus 800000 --host-api > $O/morgan_bench.jsonl 2> $O/morgan_bench.err
TRX_NO_FP4=1 python bench.py --workload morgan --n-corpus 800000 --no-cpu-baseline >> $O/morgan_bench.jsonl 2>> $O/morgan_bench.err
python tools/bigk_probe.py 1000000 16384 > $O/bigk_probe.jsonl 2> $O/bigk_probe.err
TRX_BENCH_BACKEND=gloo TRX_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 8 --weak --steps 3 --warmup 1 > $O/c2_rehearsal_bench.jsonl 2> $O/c2_rehearsal_bench.err
bash profiles/run_profile.sh $TAG > $O/run_profile.log 2>&1
python bench_predictor.py > $O/predictor_bench.jsonl 2> $O/predictor_bench.err
python bench_predictor.py --live > $O/live_bench.jsonl 2> $O/live_bench.err
bash profiles/run_profile_predictor.sh $TAG > $O/run_profile_predictor.log 2>&1
bash tools/prof_train.sh 512 160 > $O/prof_train_160.log 2>&1
python tools/step_ops.py fill > $O/step_fill.txt 2>&1
python tools/attn_ab.py no_interleave=tools/ab/libtrxnn_nointerleave.so $TAG=textreact_amd/csrc/libtrxnn.so > $O/attention_ab.json 2> $O/attention_ab.err
python tools/shard_costs.py > $O/shard_costs.json 2> $O/shard_costs.err
python tools/r05/attn_f32_ab.py > $O/attention_f32.jsonl 2> $O/attention_f32.err
TRX_NN_LIB=libtrxnn_lab.so TRX_NN_ATTN_VALU=1 python tools/r05/attn_f32_ab.py >> $O/attention_f32.jsonl 2>> $O/attention_f32.err
tail -3 $O/gputest.log; grep -c "^{" $O/bench.jsonl $O/predictor_bench.jsonl $O/live_bench.jsonl $O/fingerprint_bench.jsonl $O/c2_rehearsal_bench.jsonl
