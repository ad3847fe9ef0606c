#!/bin/bash
# round 3, call 18: attn_fwd64 loop anatomy, one probe at a time (cycles per tile of the longest query blocks)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in f64_timing t_noexp t_nodma f64_timing_mfma t_nolds t_nomax t_nobar t_bare; do
  echo "== $v"
  MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_$v.so MANTIS_ATTN_FWD64=1 timeout 120 python tools/attn_fwd64_timing.py nomask 2>&1 | grep -E "q0= *(0|768|2560|2752) "
done | tee gpurun_out/attn_fwd64_anatomy.log
