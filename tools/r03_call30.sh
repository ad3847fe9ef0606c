#!/bin/bash
# round 3, call 30: ring16 (8 waves) with s_setprio 1 on waves 4-7 -- same-box per-shape timing + step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/tools/_bin/libmantis_setprio.so
MANTIS_HIP_LIB=$L timeout 900 python tools/gpu_selftest.py gemm > gpurun_out/selftest_gemm_setprio.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_gemm_setprio.log
tail -2 gpurun_out/selftest_gemm_setprio.log; grep FAIL gpurun_out/selftest_gemm_setprio.log | head -5
echo "== default"; GEMM_BENCH_VENDOR=0 timeout 600 python tools/gemm_vs_vendor.py 10 2>&1 | grep "^[a-z_]* *[NT][NT] " | awk '{print $1, $2, $3, "v14", $16, $17}' | tee gpurun_out/gemm_ab_default.txt
echo "== setprio"; GEMM_BENCH_VENDOR=0 MANTIS_HIP_LIB=$L timeout 600 python tools/gemm_vs_vendor.py 10 2>&1 | grep "^[a-z_]* *[NT][NT] " | awk '{print $1, $2, $3, "v14", $16, $17}' | tee gpurun_out/gemm_ab_setprio.txt
for i in 1 2; do
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
MANTIS_HIP_LIB=$L timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setprio', d['value'], d['ms_per_step'])"
done
