#!/bin/bash
# Refreshes the round's evidence in ONE gpurun call:  gpurun --timeout 900 -- 'bash tools/profile_round.sh r01'
#   gpurun_out/bench_1gpu.json        default bench.py run (3 steps, CPU baseline, accuracy)
#   gpurun_out/prof_kt                rocprofv3 --kernel-trace --stats of the same command (CPU leg skipped)
#   gpurun_out/prof_{fetch,write,sq1,sq2,sq3}   one --pmc pass each (never combined with a trace domain)
# then tools/summarize_pmc.py turns them into profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_hbm.json.
set -eo pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$O/bench_1gpu.json" 2> "$O/bench_1gpu.err"
echo "bench done"
rm -rf "$O"/prof_kt "$O"/prof_fetch "$O"/prof_write "$O"/prof_sq1 "$O"/prof_sq2 "$O"/prof_sq3
rocprofv3 --kernel-trace --stats -d "$O/prof_kt" -o kt --output-format csv -- python3 "$R/bench.py" --cpu-seconds 0 > "$O/prof_kt.log" 2>&1
echo "kernel trace done"
B="python3 $R/bench.py --cpu-seconds 0 --steps 1 --warmup 0"
rocprofv3 --pmc FETCH_SIZE -d "$O/prof_fetch" -o p --output-format csv -- $B > "$O/prof_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$O/prof_write" -o p --output-format csv -- $B > "$O/prof_write.log" 2>&1
echo "hbm passes done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d "$O/prof_sq1" -o p --output-format csv -- $B > "$O/prof_sq1.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU -d "$O/prof_sq2" -o p --output-format csv -- $B > "$O/prof_sq2.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$O/prof_sq3" -o p --output-format csv -- $B > "$O/prof_sq3.log" 2>&1
echo "sq passes done"
CH=$(python3 -c "import json,sys; print(json.load(open('$O/bench_1gpu.json'))['roofline']['hbm']['chunks_per_tile'])")
cd "$R" && python3 tools/summarize_pmc.py --tag "$TAG" --kernel-trace "$O/prof_kt" --pmc "$O/prof_fetch" "$O/prof_write" "$O/prof_sq1" "$O/prof_sq2" "$O/prof_sq3" --chunks "$CH" --out-dir "$O/profiles"
cp "$O/bench_1gpu.json" "$O/profiles/${TAG}_bench_1gpu.json"
tail -c 600 "$O/bench_1gpu.json"
