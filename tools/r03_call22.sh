#!/bin/bash
# round 3, call 22: full GPU suite with attn_fwd64 as the default long-sequence forward and the host-side key-mask drop, then the headline bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err
tail -1 gpurun_out/bench_headline.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d.get('cpu_baseline'))"
MANTIS_ATTN_FWD64=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_headline_fwd32.json 2> gpurun_out/bench_headline_fwd32.err
tail -1 gpurun_out/bench_headline_fwd32.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd64 off:', d['value'], d['ms_per_step'])"
