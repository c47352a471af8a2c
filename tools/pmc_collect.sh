#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py's dominant kernel, one counter group per pass (separate runs,
# --kernel-trace only).  Usage (on the GPU box, from the repo root): tools/pmc_collect.sh <tag> [bench.py args ...]
# Output: gpurun_out/pmc_<tag>_<group>/ ... and gpurun_out/pmc_<tag>_summary.json (via tools/pmc_summarise.py)
set -u
tag=$1; shift
export TMPDIR=/tmp
export HK_NO_FIRST_PROCESS_PROBE=1   # profiles of the bench process alone
root=$(pwd)
groups=("GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE")
i=0
for g in "${groups[@]}"; do
  out=$root/gpurun_out/pmc_${tag}_$i
  rm -rf "$out"
  (cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-trace -d "$out" -o run --output-format csv -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-nan-variant --no-other-configs --no-power-probe "$@" > "$out.log" 2>&1)
  i=$((i+1))
done
python3 "$root/tools/pmc_summarise.py" "$tag" "$@" > "$root/gpurun_out/pmc_${tag}_summary.json"
cat "$root/gpurun_out/pmc_${tag}_summary.json"
