#!/bin/bash
# round 3, call 23: same-box A/B of the grad-norm pass on a side stream inside the backward (MANTIS_NORM_OVERLAP=1), now that streams
# have their own hardware queues (GPU_MAX_HW_QUEUES=8)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
  for ov in 0 1; do
    MANTIS_NORM_OVERLAP=$ov timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap=$ov', d['value'], d['ms_per_step'], d.get('ms_training_step'), d.get('ms_optimizer'))"
  done
done | tee gpurun_out/norm_overlap_ab.log
