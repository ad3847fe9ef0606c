#!/bin/bash
# round 3, GPU call 8: dW GEMMs on a side stream (A/B), final-ish validation
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/gpu_selftest.py dw_side full_width_vs ) > $O/selftest_side.log 2>&1
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_main.json 2> $O/bench_main.err
( MANTIS_DW_STREAM=1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_side.json 2> $O/bench_side.err
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_main2.json 2> $O/bench_main2.err
( MANTIS_DW_STREAM=1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_side2.json 2> $O/bench_side2.err
ls -la $O
