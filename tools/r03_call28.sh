#!/bin/bash
# round 3, call 28: the next batch's frozen tower beside clip + AdamW, re-measured now that streams have their own hardware queues
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
run() { timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d.get('ms_training_step'), d.get('ms_optimizer'))"; }
for i in 1 2; do
  run
  run --prefetch --adam-cus 0
  run --prefetch --adam-cus 192
done | tee gpurun_out/prefetch_ab.log
