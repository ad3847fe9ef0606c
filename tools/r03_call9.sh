#!/bin/bash
# round 3, GPU call 9: remainder tiles split into K parts over several sub-rounds (dX of gate|up): parity, determinism, timing, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python tools/gpu_selftest.py ksplit fullsize_gemm fullsize_linear ) > $O/selftest_split.log 2>&1
( GEMM_BENCH_VENDOR=0 timeout 500 python tools/gemm_vs_vendor.py 10 ) > $O/gemm_variants.log 2>&1
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_a.json 2> $O/bench_a.err
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_b.json 2> $O/bench_b.err
ls -la $O
