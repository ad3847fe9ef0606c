#!/bin/bash
# round 3, call 15: same-box A/B of the heavy-first XCD walk (probe library = head-major), with / without key mask
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for m in mask nomask; do
  echo "heavy-first $m"; timeout 300 python tools/attn_bench.py 20 $m 2>&1 | grep attn
  echo "head-major $m"; MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_headmajor.so timeout 300 python tools/attn_bench.py 20 $m 2>&1 | grep attn
done | tee gpurun_out/attn_bench_ab.log
