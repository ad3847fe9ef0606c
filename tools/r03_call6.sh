#!/bin/bash
# round 3, GPU call 6: attention forward fragment-prefetch depth A/B; PMC passes + kernel trace of the headline step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
for d in 3 4 5; do ( MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_fd$d.so timeout 200 python tools/attn_fwd_bench.py ) > $O/attn_fwd_fd$d.log 2>&1; done
( timeout 200 python tools/attn_fwd_bench.py ) > $O/attn_fwd_fd6.log 2>&1
( MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_fd3.so timeout 200 python tools/attn_fwd_bench.py ) > $O/attn_fwd_fd3b.log 2>&1
( timeout 300 python tools/gpu_selftest.py attn_fwd fullsize_attn_causal ) > $O/selftest_attn.log 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer"
( timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -o p -- $B ) > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
( timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -o p -- $B ) > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1
( timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -o p -- $B ) > $GRAFT_REPO_ROOT/$O/pmc_sq.log 2>&1
( timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline ) > $GRAFT_REPO_ROOT/$O/prof_c.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_step_report.py --fetch $O/pmc_fetch --write $O/pmc_write --sq $O/pmc_sq --out $O/r03_pmc_step.json > $O/pmc_report.log 2>&1
f=$(find $O/prof_c -name "p_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py $f > $O/prof_c_stats.md 2>&1
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +20M -delete
ls -la $O
