#!/bin/bash
# Refreshes the round's evidence in ONE gpurun call:  gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# Everything lands under gpurun_out/<tag>/ ; tools/collect_profiles.py <tag> then copies the summaries into profiles/.
#   bench_1gpu.json      default bench.py run (live PMC, CPU baseline, accuracy) + pmc_hbm.json (the counters as a summary)
#                        + pmc_hbm_c2_pmc_hbm.json (FETCH_SIZE / WRITE_SIZE of the 1024-spp launch, BASELINE configs[2])
#   kt/                  rocprofv3 --kernel-trace --stats of the same command (no PMC in that run, CPU leg skipped)
#   configs.jsonl        every BASELINE configuration on one GPU (tools/run_configs.py)
#   mutation_sweep.jsonl tools/mutation_sweep.py
#   band_balance.jsonl   tools/band_balance_probe.py (kernel time of every band of the multi-device splits, on one GPU)
#   pmc_tor/ pmc_x64/ pmc_sky/   detailed SQ / TCP counter passes (one group per pass, never with a trace domain)
#   phase_*.log          per-phase shader-clock shares from the diagnostic build (libpt_phase.so)
#   blockprof_*          execution counters of the instrumented code object (libpt_blockprof.so + lib/blockprof/pt_bp.hsaco)
set -eo pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 10 --warmup 2 --save-pmc "$O/pmc_hbm.json" > "$O/bench_1gpu.json" 2> "$O/bench_1gpu.err"
echo "bench done"
rocprofv3 --kernel-trace --stats -d "$O/kt" -o kt --output-format csv -- python3 "$R/bench.py" --cpu-seconds 0 --pmc off --no-configs3 --no-extra-legs > "$O/kt.log" 2>&1
echo "kernel trace done"
python3 "$R/tools/run_configs.py" > "$O/configs.jsonl" 2> "$O/configs.err"
echo "configs done"
python3 "$R/tools/mutation_sweep.py" > "$O/mutation_sweep.jsonl" 2> "$O/mutation.err"
echo "mutation sweep done"
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES"
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU"
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum")
python3 "$R/tools/make_replicated_scene.py" --instances 64 --out-dir "$O/x64scene" > /dev/null
python3 "$R/tools/make_open_scene.py" --out-dir "$O/openscene" > /dev/null
PTR="$R/path-tracing_amd/bin/pt_render"
FRAME="--W 1920 --H 1080 -RPP 256 -MRR 8 -ERR -1 -SEED 42 -BENCH_STEPS 1 -BENCH_WARMUP 1"
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g -d "$O/pmc_tor/p$i" -o p --output-format csv -- "$PTR" $FRAME -MODEL_PATH "$R/models/" > "$O/pmc_tor_p$i.log" 2>&1
  rocprofv3 --pmc $g -d "$O/pmc_x64/p$i" -o p --output-format csv -- "$PTR" $FRAME -MODEL_PATH "$O/x64scene/" -MODEL_NAME TorX64.obj > "$O/pmc_x64_p$i.log" 2>&1
  rocprofv3 --pmc $g -d "$O/pmc_sky/p$i" -o p --output-format csv -- "$PTR" $FRAME -MODEL_PATH "$O/openscene/" -MODEL_NAME TorOpen.obj -SKYBOX "$O/openscene/sky.bmp" > "$O/pmc_sky_p$i.log" 2>&1
done
# kernel trace of the open scene under its sky (the skybox instantiation with path regeneration), the C++ front end, 5 frames
rocprofv3 --kernel-trace --stats -d "$O/kt_sky" -o kt --output-format csv -- "$PTR" --W 1920 --H 1080 -RPP 256 -MRR 8 -ERR -1 -SEED 42 -BENCH_STEPS 5 -BENCH_WARMUP 1 -MODEL_PATH "$O/openscene/" -MODEL_NAME TorOpen.obj -SKYBOX "$O/openscene/sky.bmp" > "$O/kt_sky.log" 2>&1
# the open-scene probe (live rays per wave-segment by -MRR) and the triangle-count sweep across the small / big switch
python3 "$R/tools/open_scene_probe.py" --spp 64 > "$O/open_scene_probe.jsonl" 2> "$O/open_scene_probe.err"
python3 "$R/tools/t_sweep.py" > "$O/t_sweep.jsonl" 2> "$O/t_sweep.err"
# what every device of a 2 / 4 / 8-device node would have to render: contiguous row bands against the interleaved split, band by band on this GPU
python3 "$R/tools/band_balance_probe.py" 2> "$O/band_balance.err" | grep frame > "$O/band_balance.jsonl"
echo "pmc detail done"
PT_HIP_LIB=$R/path-tracing_amd/lib/libpt_phase.so python3 "$R/tools/tor_probe.py" > "$O/phase_tor.log" 2>&1
PT_HIP_LIB=$R/path-tracing_amd/lib/libpt_phase.so python3 "$R/tools/c5_probe.py" 64,195 > "$O/phase_x64_x195.log" 2>&1
echo "phase timers done"
# dynamic instruction profile (tools/asm_profile.py): instrumented code object, execution counters per straight-line run
PT_BLOCKPROF_OUT=$O/blockprof_tor python3 "$R/tools/blockprof_run.py" tor 8 > "$O/blockprof_tor.log" 2>&1
PT_BLOCKPROF_OUT=$O/blockprof_x64 python3 "$R/tools/blockprof_run.py" x64 4 > "$O/blockprof_x64.log" 2>&1
echo "block profile done"
# where a fresh process spends its time outside the kernels (runtime start-up, first calls, teardown)
for i in 1 2 3; do ( time "$R/path-tracing_amd/lib/tools/hip_startup_probe" ) >> "$O/hip_startup_probe.txt" 2>&1; done
for i in 1 2 3; do "$PTR" --W 1920 --H 1080 -RPP 256 -MRR 8 -ERR -1 -UPDATE 0 -QUIET 1 -MODEL_PATH "$R/models/" -OUT /tmp/e2e.bmp -TIMING 1 -FASTEXIT 1 -T0_NS $(date +%s%N) 2>> "$O/e2e_cold_phases.txt" > /dev/null; done
echo "start-up probe done"
tail -c 400 "$O/bench_1gpu.json"
