#!/bin/bash
# Several rocprofv3 --pmc passes (one counter group each, never with a trace domain) over one command; prints the
# integrator's counters.   bash tools/pmc_passes.sh <outdir-name> -- python3 tools/c5_probe.py 64
set -eo pipefail
NAME=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$NAME
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $group -d "$O/p$i" -o p --output-format csv -- "$@" > "$O/p$i.log" 2>&1 || { tail -5 "$O/p$i.log"; exit 1; }
  grep integrate "$O/p$i/p_counter_collection.csv" | awk -F'","|",|,"' '{print $(NF-3), $(NF-2)}' | sed 's/"//g'
done <<GROUPS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
GROUPS
