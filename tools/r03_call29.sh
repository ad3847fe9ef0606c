#!/bin/bash
# round 3, call 29: AdamW with streaming (nt) loads and stores -- parity of the optimizer checks, same-box step timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/tools/_bin/libmantis_adamw_nt.so
MANTIS_HIP_LIB=$L timeout 600 python tools/gpu_selftest.py adam optim clip > gpurun_out/selftest_adamw_nt.log 2>&1
tail -2 gpurun_out/selftest_adamw_nt.log; grep FAIL gpurun_out/selftest_adamw_nt.log | head
run() { timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('ms_training_step'), d.get('ms_optimizer'))"; }
for i in 1 2; do
  run default
  MANTIS_HIP_LIB=$L run adamw_nt
done | tee gpurun_out/adamw_nt_ab.log
