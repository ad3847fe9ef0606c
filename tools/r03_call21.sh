#!/bin/bash
# round 3, call 21: attn_fwd64 with the overlapped prologue, the LDS-staged epilogue, prefetched liveness words and spread DMA pieces
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
MANTIS_ATTN_FWD64=1 timeout 900 python tools/gpu_selftest.py attn > gpurun_out/selftest_fwd64.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_fwd64.log
tail -2 gpurun_out/selftest_fwd64.log; grep FAIL gpurun_out/selftest_fwd64.log | head -8
echo base; timeout 300 python tools/attn_fwd_bench.py 2>&1 | grep "hd 128"
echo fwd64; MANTIS_ATTN_FWD64=1 timeout 300 python tools/attn_fwd_bench.py 2>&1 | grep "hd 128"
for m in mask nomask; do echo "base $m"; timeout 300 python tools/attn_bench.py 20 $m 2>&1 | grep attn; echo "fwd64 $m"; MANTIS_ATTN_FWD64=1 timeout 300 python tools/attn_bench.py 20 $m 2>&1 | grep attn; done
for v in f64_timing; do
  echo "== $v"
  MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_$v.so MANTIS_ATTN_FWD64=1 timeout 120 python tools/attn_fwd64_timing.py nomask 2>&1 | grep -E "q0= *(0|768|2560|2752) "
  MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_$v.so MANTIS_ATTN_FWD64=1 timeout 120 python tools/attn_fwd64_timing.py mask 2>&1 | grep -E "q0= *(0|768|2560|2752) "
done
