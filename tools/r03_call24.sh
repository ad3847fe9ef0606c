#!/bin/bash
# round 3, call 24: end-of-round measurements -- kernel trace of the headline step, the three configurations' bench lines
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_d -o p -- python $R/bench.py --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline > $R/gpurun_out/prof_d.log 2>&1
cd $R
db=$(find gpurun_out/prof_d -name "*.db" | head -1)
python tools/rocpd_stats.py $db > gpurun_out/prof_d_stats.md
rm -rf gpurun_out/prof_d
head -30 gpurun_out/prof_d_stats.md
timeout 900 python bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err
timeout 900 python bench.py --config mantis_8b_idefics2 --no-cpu-baseline > gpurun_out/bench_idefics2.json 2> gpurun_out/bench_idefics2.err
timeout 900 python bench.py --config qwen2_vl_7b --no-cpu-baseline > gpurun_out/bench_qwen2vl.json 2> gpurun_out/bench_qwen2vl.err
for f in headline idefics2 qwen2vl; do tail -1 gpurun_out/bench_$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'])"; done
