#!/bin/bash
# round 3, call 16: s_memtime anatomy of attn_fwd64 (prologue / loop / epilogue per wave)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in f64_timing f64_timing_mfma; do
  for m in nomask mask; do
    echo "== $v"
    MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_$v.so MANTIS_ATTN_FWD64=1 timeout 300 python tools/attn_fwd64_timing.py $m 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/attn_fwd64_timing.log
