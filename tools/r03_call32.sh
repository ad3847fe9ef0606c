#!/bin/bash
# round 3, call 32: AdamW over the packed fp32 state -- parity (kernel check, optimizer end-to-end vs torch, norm overlap, DP checks), step timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gpu_selftest.py optim optimizer norm_overlap adam dp_ grad > gpurun_out/selftest_optim.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_optim.log
tail -3 gpurun_out/selftest_optim.log; grep -E "FAIL|PASS" gpurun_out/selftest_optim.log | head -20
for i in 1 2; do
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('packed', d['value'], d['ms_per_step'], d.get('ms_training_step'), d.get('ms_optimizer'))"
done
