#!/bin/bash
# round 3, call 34: optimizer pass vs the relative placement of its three fp32 state arrays (one allocation, starts skewed by MANTIS_ADAM_SKEW bytes)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
run() { timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skew $MANTIS_ADAM_SKEW', d['value'], d['ms_per_step'], d.get('ms_training_step'), d.get('ms_optimizer'))"; }
for sk in 0 4352 1052928 33558784 0 1052928; do MANTIS_ADAM_SKEW=$sk run; done | tee gpurun_out/adam_skew.log
