#!/bin/bash
# round 3, GPU call 5: 8-wave pair epilogue fix, perceiver cross attention; full suite; benches of the three configs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( timeout 500 python tools/gpu_selftest.py fused attn_cross idefics2 model_step_cfg1 ) > $O/selftest_a.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > $O/bench_a.json 2> $O/bench_a.err
( timeout 500 python bench.py --config mantis_8b_idefics2 --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_idefics2.json 2> $O/bench_idefics2.err
( timeout 500 python bench.py --config qwen2_vl_7b --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_qwen.json 2> $O/bench_qwen.err
( timeout 500 python bench.py --config qwen2_vl_7b --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline ) > $O/bench_qwen_bf16.json 2> $O/bench_qwen_bf16.err
ls -la $O
