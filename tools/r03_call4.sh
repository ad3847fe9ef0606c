#!/bin/bash
# round 3, GPU call 4: fused SwiGLU / RoPE epilogues, per-shape ring choice, device-side NaViT preparation; full suite; bench + profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python tools/gpu_selftest.py fused navit fullsize_linear ) > $O/selftest_fused.log 2>&1
( timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_fused.json 2> $O/bench_fused.err
( MANTIS_NO_FUSE=1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_unfused.json 2> $O/bench_unfused.err
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_fused2.json 2> $O/bench_fused2.err
( timeout 500 python bench.py --config mantis_8b_idefics2 --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_idefics2.json 2> $O/bench_idefics2.err
cd /tmp
( timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_b -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline ) > $GRAFT_REPO_ROOT/$O/prof_b.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_b -name "p_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f > $O/prof_b_stats.md 2>&1
find $O -name "*.db" -size +30M -delete
ls -la $O
