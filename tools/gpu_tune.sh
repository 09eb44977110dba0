#!/bin/bash
# quick A/B of tuning knobs on the GPU box (bench only, no CPU baseline)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -n 5 gpurun_out/pytest_gpu.log
for R in 0; do
  if [ "$R" = "0" ]; then unset SLAM2D_SWEEP_R; else export SLAM2D_SWEEP_R=$R; fi
  echo "== SWEEP_R=$R"
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: v['avg_us'] for k, v in d['stages_probe'].items()})
"
done
unset SLAM2D_SWEEP_R
echo "== ref2level"
timeout 300 python bench.py --workload ref2level --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: v['avg_us'] for k, v in d['stages_probe'].items()})
"
echo "== forced dist (nccl, 1 rank)"
SLAM2D_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -n 2 | cut -c1-400
echo "== torchrun 1 rank"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -n 2 | cut -c1-300
echo "== config5"
timeout 300 python bench.py --workload config5 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: v['avg_us'] for k, v in d['stages_probe'].items()})
"
