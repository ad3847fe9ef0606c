#!/bin/bash
# round 3, call 17: which torch elementwise kernels run inside the headline step, and where
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_t -o t -- python $R/bench.py --steps 2 --warmup 1 --no-kernel-timer --no-cpu-baseline > $R/gpurun_out/prof_t.log 2>&1
cd $R
db=$(find gpurun_out/prof_t -name "*.db" | head -1)
python tools/rocpd_torch_kernels.py $db > gpurun_out/torch_kernels.txt
python tools/rocpd_stats.py $db > gpurun_out/prof_t_stats.md
rm -rf gpurun_out/prof_t
head -80 gpurun_out/torch_kernels.txt
