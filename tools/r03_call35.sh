#!/bin/bash
# round 3, call 35: final state -- full GPU suite, the three configurations' bench lines
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err
timeout 900 python bench.py --config mantis_8b_idefics2 --no-cpu-baseline > gpurun_out/bench_idefics2.json 2> gpurun_out/bench_idefics2.err
timeout 900 python bench.py --config qwen2_vl_7b --no-cpu-baseline > gpurun_out/bench_qwen2vl.json 2> gpurun_out/bench_qwen2vl.err
for f in headline idefics2 qwen2vl; do tail -1 gpurun_out/bench_$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
