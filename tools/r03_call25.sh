#!/bin/bash
# round 3, call 25: ring16 with the two waves of a SIMD de-phased (loads of one under the MFMAs of the other): parity of the GEMM checks,
# per-shape timing against the default build on the same box, headline step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
L=$PWD/tools/_bin/libmantis_dephase.so
MANTIS_HIP_LIB=$L timeout 900 python tools/gpu_selftest.py gemm > gpurun_out/selftest_gemm_dephase.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_gemm_dephase.log
tail -2 gpurun_out/selftest_gemm_dephase.log; grep FAIL gpurun_out/selftest_gemm_dephase.log | head -5
echo "== default"; GEMM_BENCH_VENDOR=0 timeout 600 python tools/gemm_vs_vendor.py 10 2>&1 | grep "|" | tee gpurun_out/gemm_ab_default.md
echo "== dephase"; GEMM_BENCH_VENDOR=0 MANTIS_HIP_LIB=$L timeout 600 python tools/gemm_vs_vendor.py 10 2>&1 | grep "|" | tee gpurun_out/gemm_ab_dephase.md
for i in 1 2; do
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
MANTIS_HIP_LIB=$L timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dephase', d['value'], d['ms_per_step'])"
done
