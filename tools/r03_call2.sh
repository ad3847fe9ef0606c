#!/bin/bash
# round 3, GPU call 2: ring16 GEMM (4 waves x 128x128 of 16x16x32) -- parity, A/B vs the 8-wave kernel and the vendor, step bench; HW-queue experiment for the DP stream
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/gpu_selftest.py gemm linear ) > $O/selftest_gemm.log 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( GEMM_BENCH_VENDOR=1 timeout 500 python tools/gemm_vs_vendor.py 10 ) > $O/gemm_vs_vendor.log 2>&1
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_r16.json 2> $O/bench_r16.err
( MANTIS_GEMM_RING=12 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_r12.json 2> $O/bench_r12.err
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_r16b.json 2> $O/bench_r16b.err
# DP world-1 with more hardware queues
( GPU_MAX_HW_QUEUES=8 MANTIS_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_dpforce_q8.json 2> $O/bench_dpforce_q8.err
cd /tmp
( GPU_MAX_HW_QUEUES=8 MANTIS_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_dpforce_q8 -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline ) > $GRAFT_REPO_ROOT/$O/prof_dpforce_q8.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_dpforce_q8 -name "f_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f > $O/prof_dpforce_q8_stats.md 2>&1
ls -la $O
