#!/bin/bash
# Multi-GPU sweep of the headline step on ONE node: bench.py --gpus N for N in $GPUS x gradient-exchange algorithm x RCCL channel count x
# the GEMM planner's CU budget, then the scaling efficiency of every combination against its own N = 1 line.  Nothing here has run on more
# than one GPU yet (no 8-GPU node was available to the builder or, so far, to the driver): the first hour on such a node should be this script.
#
#   tools/scale_sweep.sh                                   # defaults below
#   GPUS="1 2 4 8" ALGOS="allreduce rs_ag" CHANNELS="default 8 16 32" GEMM_CUS="0 240 224" STEPS=10 WARMUP=3 tools/scale_sweep.sh
#
# Knobs swept (see mantis_amd/dp.py and csrc/gemm.hip):
#   MANTIS_DP_ALGO       allreduce | rs_ag (reduce-scatter + all-gather per bucket: both phases are direct exchanges on the fully connected
#                        xGMI node, all 7 links at once; a ring all-reduce is per-link bound)
#   NCCL_MAX_NCHANNELS   RCCL channels = workgroups = CUs taken from the concurrent GEMM ("default": RCCL's choice)
#   MANTIS_GEMM_CUS      CU budget the GEMM tile scheduler plans rounds and K-splits for (0 = all CUs of the device): with C channels busy,
#                        planning for 256 - C keeps a ring GEMM's last round from waiting for CUs RCCL holds
# Output: gpurun_out/scale_sweep.jsonl (one bench line per run, plus the knobs) and a table on stdout.
cd "$(dirname "$0")/.."
O=${OUT:-gpurun_out}; mkdir -p $O
GPUS=${GPUS:-"1 2 4 8"}; ALGOS=${ALGOS:-"allreduce rs_ag"}; CHANNELS=${CHANNELS:-"default 16 32"}; GEMM_CUS=${GEMM_CUS:-"0 240"}
STEPS=${STEPS:-10}; WARMUP=${WARMUP:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
# a measured data-parallel number must not come from a silently serialised exchange (round-5 advisor finding): the reducer's overlap probe is an
# ERROR here, not a warning; the probe's verdict is also on every bench line (dp.overlap_probe)
export MANTIS_DP_REQUIRE_OVERLAP=${MANTIS_DP_REQUIRE_OVERLAP:-1}
STAGES=${STAGES:-"finetune pretrain"}      # pretrain = projector only (43 MB of gradients, pretrain_mllava.sh:186): the trivially-communicating control
: > $O/scale_sweep.jsonl
for stage in $STAGES; do for algo in $ALGOS; do for ch in $CHANNELS; do for cus in $GEMM_CUS; do for n in $GPUS; do
    # the projector-only control needs one knob setting only
    if [ "$stage" = pretrain ] && { [ "$algo" != "${ALGOS%% *}" ] || [ "$ch" != "${CHANNELS%% *}" ] || [ "$cus" != "${GEMM_CUS%% *}" ]; }; then continue; fi
    # at N = 1 there is no exchange: one line per CU budget is enough
    if [ "$n" = 1 ] && { [ "$algo" != "${ALGOS%% *}" ] || [ "$ch" != "${CHANNELS%% *}" ]; }; then continue; fi
    envs="MANTIS_DP_ALGO=$algo MANTIS_GEMM_CUS=$cus"
    [ "$ch" != default ] && envs="$envs NCCL_MAX_NCHANNELS=$ch NCCL_MIN_NCHANNELS=$ch"
    line=$(env $envs timeout 900 python bench.py --gpus $n --stage $stage --steps $STEPS --warmup $WARMUP --no-cpu-baseline 2>$O/scale_sweep.err | tail -1)
    [ -z "$line" ] && { echo "FAILED: $envs --gpus $n (see $O/scale_sweep.err)" >&2; continue; }
    python - "$algo" "$ch" "$cus" "$line" "$stage" >> $O/scale_sweep.jsonl <<'PY'
import json, sys
algo, ch, cus, line, stage = sys.argv[1:6]
d = json.loads(line)
assert (d.get("dp") or {}).get("world_size_seen", d["n_gpus"]) == d["n_gpus"], "the communicator did not see n_gpus ranks"
print(json.dumps(dict(stage=stage, algo=algo, channels=ch, gemm_cus=int(cus), n_gpus=d["n_gpus"], value=d["value"], ms_per_step=d["ms_per_step"],
                      ms_training_step=d["ms_training_step"], dp=d.get("dp"), frac=(d.get("roofline") or {}).get("frac"))))
PY
done; done; done; done; done
python - $O/scale_sweep.jsonl <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = {(r.get("stage"), r["gemm_cus"]): r["value"] for r in rows if r["n_gpus"] == 1}
print("| stage | algo | channels | GEMM CU budget | GPUs | samples/s | ms/step | exposed comm ms (median) | efficiency vs N=1 |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    b = base.get((r.get("stage"), r["gemm_cus"])) or max([v for (st, _), v in base.items() if st == r.get("stage")] or [0]) or None
    eff = "" if not b else f"{r['value'] / (r['n_gpus'] * b):.3f}"
    ex = (r.get("dp") or {}).get("exposed_comm_ms_median")
    print(f"| {r.get('stage')} | {r['algo']} | {r['channels']} | {r['gemm_cus'] or 'all'} | {r['n_gpus']} | {r['value']:.3f} | {r['ms_per_step']:.1f} | {'' if ex is None else ex} | {eff} |")
PY
