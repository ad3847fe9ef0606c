#!/bin/bash
# round 3, call 11: attn_fwd64 (64 query rows per wave) -- parity of every attention check with the switch on, then A/B timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
MANTIS_ATTN_FWD64=1 timeout 900 python tools/gpu_selftest.py attn > gpurun_out/selftest_fwd64.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_fwd64.log
cp gpurun_out/selftest.json gpurun_out/selftest_fwd64.json 2>/dev/null
tail -5 gpurun_out/selftest_fwd64.log
grep FAIL gpurun_out/selftest_fwd64.log | head -20
timeout 300 python tools/attn_fwd_bench.py > gpurun_out/attn_fwd_bench_base.log 2>&1
MANTIS_ATTN_FWD64=1 timeout 300 python tools/attn_fwd_bench.py > gpurun_out/attn_fwd_bench_fwd64.log 2>&1
echo base; grep "hd 128" gpurun_out/attn_fwd_bench_base.log
echo fwd64; grep "hd 128" gpurun_out/attn_fwd_bench_fwd64.log
