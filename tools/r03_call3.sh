#!/bin/bash
# round 3, GPU call 3: ring16 with 4 and 8 waves vs the 32x32x16 ring kernel, per shape; new cfg4/cfg5-shape oracle checks; full suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python tools/gpu_selftest.py gemm linear ) > $O/selftest_gemm.log 2>&1
( GEMM_BENCH_VENDOR=1 timeout 600 python tools/gemm_vs_vendor.py 10 ) > $O/gemm_vs_vendor.log 2>&1
( timeout 900 python tools/gpu_selftest.py fullsize_attn_cfg fullsize_fp8 ) > $O/selftest_cfg45.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( MANTIS_GEMM_RING=14 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_r14.json 2> $O/bench_r14.err
( MANTIS_GEMM_RING=12 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_r12.json 2> $O/bench_r12.err
( MANTIS_GEMM_RING=13 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_r13.json 2> $O/bench_r13.err
ls -la $O
