#!/bin/bash
# round 3, call 13: heavy-first (tile-major) XCD walk in the causal attention kernels: parity of every attention check, then fwd / bwd timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gpu_selftest.py attn > gpurun_out/selftest_attn.log 2>&1
echo "selftest rc=$?" >> gpurun_out/selftest_attn.log
tail -3 gpurun_out/selftest_attn.log; grep FAIL gpurun_out/selftest_attn.log | head
for i in 1 2; do timeout 300 python tools/attn_bench.py 20; done 2>&1 | tee gpurun_out/attn_bench_heavy.log
MANTIS_ATTN_FWD64=1 timeout 300 python tools/attn_bench.py 20 2>&1 | tee gpurun_out/attn_bench_heavy_fwd64.log
timeout 300 python tools/attn_fwd_bench.py 2>&1 | grep "hd 128" | tee gpurun_out/attn_fwd_bench_heavy.log
MANTIS_ATTN_FWD64=1 timeout 300 python tools/attn_fwd_bench.py 2>&1 | grep "hd 128" | tee gpurun_out/attn_fwd_bench_heavy_fwd64.log
