#!/bin/bash
# Round-end evidence in ONE gpurun call: the three bench lines (with cpu_baseline) + the headline through the HF loop, the in-step per-shape GEMM table, a kernel trace of
# the headline step and the PMC passes (separate runs, --kernel-trace only beside --pmc).  Everything lands in gpurun_out/ev_*; copy what
# is worth keeping into profiles/ (names per round).  Optional: SUITE=1 runs the GPU test suite first (~7.5 min).
#   gpurun --timeout 2400 -- 'bash tools/gpu_call.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; R=$PWD; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m mantis_amd.build >/dev/null 2>&1
if [ "${SUITE:-0}" = 1 ]; then MANTIS_CHECK_REPORT_DIR=$R/$O timeout 1800 python -m pytest tests -m gpu -x -q > $O/ev_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/ev_pytest_gpu.log; fi
timeout 600 python bench.py --gemm-table $O/ev_gemm_in_step.md > $O/ev_bench_headline.json 2>$O/ev_bench_headline.err; tail -c 400 $O/ev_bench_headline.json; echo
timeout 600 python bench.py --loop hf --steps 8 --warmup 3 --no-cpu-baseline > $O/ev_bench_hf_loop.json 2>$O/ev_bench_hf_loop.err      # the same workload through transformers.Trainer.train()
timeout 600 python bench.py --config mantis_8b_idefics2 > $O/ev_bench_idefics2.json 2>$O/ev_bench_idefics2.err
timeout 600 python bench.py --stage pretrain --no-cpu-baseline > $O/ev_bench_pretrain.json 2>$O/ev_bench_pretrain.err                      # projector-only stage (pretrain_mllava.sh:186)
timeout 600 python bench.py --config mantis_8b_clip_llama3 --no-cpu-baseline > $O/ev_bench_clip.json 2>$O/ev_bench_clip.err              # the scripts' default tower (pretrain_mllava.sh:34)
MANTIS_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/ev_bench_dp_world1.json 2>$O/ev_bench_dp_world1.err   # RCCL at world size 1: the dp section
timeout 600 python bench.py --config qwen2_vl_7b --precision fp8 > $O/ev_bench_qwen2vl_fp8.json 2>$O/ev_bench_qwen2vl_fp8.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $R/$O/ev_prof -o p -- python $R/bench.py --steps 3 --warmup 2 --no-kernel-timer --no-cpu-baseline > /dev/null 2>&1
cd $R; python tools/rocpd_stats.py $(find $O/ev_prof -name "p_results.db" | head -1) > $O/ev_kernel_stats.md 2>/dev/null; rm -rf $O/ev_prof
cd /tmp; B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o p -- $B > /dev/null 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o p -- $B > /dev/null 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_sq -o p -- $B > /dev/null 2>&1
cd $R; python tools/pmc_step_report.py --fetch $O/pmc_fetch --write $O/pmc_write --sq $O/pmc_sq --out $O/ev_pmc_step.json > $O/ev_pmc_step.txt 2>&1
python tools/pmc_kernel.py $O/pmc_sq gemm > $O/ev_pmc_gemm.txt 2>&1; python tools/pmc_kernel.py $O/pmc_sq attn > $O/ev_pmc_attn.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
python - <<'PY'
import json
for n in ("headline", "hf_loop", "idefics2", "qwen2vl_fp8", "pretrain", "clip", "dp_world1"):
    try:
        d = [json.loads(l) for l in open(f"gpurun_out/ev_bench_{n}.json") if l.startswith('{"')][-1]
        print(n, d["ms_per_step"], d["ms_training_step"], d["value"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"), d.get("native_loop"))
    except Exception as e:
        print(n, "FAILED", e)
PY
