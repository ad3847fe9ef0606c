#!/bin/bash
# round 3, GPU call 1: MFMA shape probe, full GPU suite on the patched library, bench, vendor A/B, world-1 RCCL attribution
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 120 tools/_bin/mfma_shape_probe ) > $O/mfma_shape_probe.log 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( timeout 600 python bench.py --steps 10 --warmup 3 ) > $O/bench_a.json 2> $O/bench_a.err
( GEMM_BENCH_VENDOR=1 timeout 400 python tools/gemm_vs_vendor.py 10 ) > $O/gemm_vs_vendor.log 2>&1
# world-1 RCCL: (b) forced, same process, no torchrun, OMP untouched; (c) forced under torchrun (OMP_NUM_THREADS=1); (a2) plain again
( MANTIS_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_dpforce_inproc.json 2> $O/bench_dpforce_inproc.err
( MANTIS_DP_FORCE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_dpforce_torchrun.json 2> $O/bench_dpforce_torchrun.err
( OMP_NUM_THREADS=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_plain_omp1.json 2> $O/bench_plain_omp1.err
# kernel traces: plain and forced
cd /tmp
( timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_plain -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline ) > $GRAFT_REPO_ROOT/$O/prof_plain.log 2>&1
( MANTIS_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_dpforce -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline ) > $GRAFT_REPO_ROOT/$O/prof_dpforce.log 2>&1
cd $GRAFT_REPO_ROOT
for n in plain:p dpforce:f; do d=${n%%:*}; o=${n##*:}; f=$(find $O/prof_$d -name "${o}_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f > $O/prof_${d}_stats.md 2>&1; done
# keep the traces small: drop the databases (64 MiB merge limit), keep the stats
find $O -name "*.db" -size +20M -delete
ls -la $O
