#!/bin/bash
# round 3, GPU call 7: where the ring16 loop's idle MFMA cycles go (timing probes), new full-width-vs-oracle steps, fp8 bars, full suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python tools/gemm_probe_bench.py 10 ) > $O/probe_base.log 2>&1
for v in NO_DMA NO_FRAGS NO_BARRIER; do ( MANTIS_HIP_LIB=$PWD/tools/_bin/libmantis_$v.so timeout 200 python tools/gemm_probe_bench.py 10 ) > $O/probe_$v.log 2>&1; done
( timeout 200 python tools/gemm_probe_bench.py 10 ) > $O/probe_base2.log 2>&1
( timeout 900 python tools/gpu_selftest.py full_width fp8_step fp8_llava ) > $O/selftest_fw.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
ls -la $O
