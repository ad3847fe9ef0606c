#!/bin/bash
# round 3, GPU call 10: attention forward with alternating score accumulators (A/B), parity, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python tools/attn_fwd_bench.py ) > $O/attn_fwd_il1.log 2>&1
( MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_il0.so timeout 200 python tools/attn_fwd_bench.py ) > $O/attn_fwd_il0.log 2>&1
( timeout 200 python tools/attn_fwd_bench.py ) > $O/attn_fwd_il1b.log 2>&1
( MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_il0.so timeout 200 python tools/attn_fwd_bench.py ) > $O/attn_fwd_il0b.log 2>&1
( timeout 600 python tools/gpu_selftest.py attn ) > $O/selftest_attn.log 2>&1
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_a.json 2> $O/bench_a.err
ls -la $O
