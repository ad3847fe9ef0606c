#!/bin/bash
# round 3, call 26: SQ counters of the attention kernels (MFMA pipe busy, clock, wave-cycle split), 64-row and 32-row forward
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for m in 1 0; do
  cd /tmp && MANTIS_ATTN_FWD64=$m timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_attn$m -o p -- python $R/tools/attn_bench.py 5 nomask > $R/gpurun_out/pmc_attn$m.log 2>&1
  cd $R
  echo "== MANTIS_ATTN_FWD64=$m"; python tools/pmc_kernel.py gpurun_out/pmc_attn$m attn
  rm -rf gpurun_out/pmc_attn$m
done | tee gpurun_out/pmc_attn.txt
