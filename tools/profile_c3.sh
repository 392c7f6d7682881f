#!/bin/bash
# BASELINE.json configs[2] ("rocprof HBM GB/s run"): Tor.obj 1920x1080 x 1024 spp under rocprofv3.
#   gpurun --timeout 900 -- 'bash tools/profile_c3.sh r01_c3'
set -eo pipefail
TAG=${1:-r01_c3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --spp 1024 --cpu-seconds 0 --steps 1 --warmup 0"
$B > "$O/bench_c3.json" 2> "$O/bench_c3.err"
rm -rf "$O"/c3_kt "$O"/c3_fetch "$O"/c3_write "$O"/c3_sq
rocprofv3 --kernel-trace --stats -d "$O/c3_kt" -o kt --output-format csv -- $B > "$O/c3_kt.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d "$O/c3_fetch" -o p --output-format csv -- $B > "$O/c3_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$O/c3_write" -o p --output-format csv -- $B > "$O/c3_write.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES -d "$O/c3_sq" -o p --output-format csv -- $B > "$O/c3_sq.log" 2>&1
echo "pmc passes done"
CH=$(python3 -c "import json; print(json.load(open('$O/bench_c3.json'))['roofline']['hbm']['chunks_per_tile'])")
cd "$R" && python3 tools/summarize_pmc.py --tag "$TAG" --spp 1024 --kernel-trace "$O/c3_kt" --pmc "$O/c3_fetch" "$O/c3_write" "$O/c3_sq" --chunks "$CH" --out-dir "$O/profiles"
cut -c1-300 "$O/bench_c3.json"
