#!/bin/bash
# round 3, call 12: attn_fwd64 after the permlane hazard fix + 4-slot ring: parity, A/B timing, timing probes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
MANTIS_ATTN_FWD64=1 timeout 900 python tools/gpu_selftest.py attn > gpurun_out/selftest_fwd64.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_fwd64.log
cp gpurun_out/selftest.json gpurun_out/selftest_fwd64.json 2>/dev/null
tail -3 gpurun_out/selftest_fwd64.log
grep FAIL gpurun_out/selftest_fwd64.log | head -12
timeout 300 python tools/attn_fwd_bench.py > gpurun_out/attn_fwd_bench_base.log 2>&1
echo base; grep "hd 128" gpurun_out/attn_fwd_bench_base.log
for v in hip f64_novm f64_noexp f64_nodma f64_mfma; do
  lib=tools/_bin/libmantis_$v.so; [ $v = hip ] && lib=mantis_amd/libmantis_hip.so
  MANTIS_HIP_LIB=$PWD/$lib MANTIS_ATTN_FWD64=1 timeout 300 python tools/attn_fwd_bench.py > gpurun_out/attn_fwd_bench_$v.log 2>&1
  echo fwd64 $v; grep "hd 128" gpurun_out/attn_fwd_bench_$v.log
done
