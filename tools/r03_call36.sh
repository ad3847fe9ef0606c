#!/bin/bash
# round 3, call 36: PMC passes of the headline step at the final build (-> profiles/r03_pmc_step.json, which bench.py's roofline.traffic reads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer"
( timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o p -- $B ) > $R/$O/pmc_fetch.log 2>&1
( timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o p -- $B ) > $R/$O/pmc_write.log 2>&1
( timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_sq -o p -- $B ) > $R/$O/pmc_sq.log 2>&1
cd $R
python tools/pmc_step_report.py --fetch $O/pmc_fetch --write $O/pmc_write --sq $O/pmc_sq --out $O/r03_pmc_step.json > $O/pmc_report.log 2>&1
python tools/pmc_kernel.py $O/pmc_sq attn > $O/pmc_attn_in_step.txt 2>&1
python tools/pmc_kernel.py $O/pmc_sq gemm > $O/pmc_gemm_in_step.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
tail -5 $O/pmc_report.log; cat $O/pmc_attn_in_step.txt; head -12 $O/pmc_gemm_in_step.txt
