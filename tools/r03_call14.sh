#!/bin/bash
# round 3, call 14: key-mask word prefetched one stage ahead (fwd, dQ) on top of the heavy-first XCD walk: parity, kernel timing, headline step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gpu_selftest.py attn > gpurun_out/selftest_attn.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_attn.log
tail -3 gpurun_out/selftest_attn.log; grep FAIL gpurun_out/selftest_attn.log | head
for i in 1 2; do timeout 300 python tools/attn_bench.py 20; done 2>&1 | grep attn | tee gpurun_out/attn_bench_kmpref.log
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/bench_call14.json 2> gpurun_out/bench_call14.err
cat gpurun_out/bench_call14.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline'))"
