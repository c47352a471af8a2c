#!/bin/bash
# kernel time of the in-painting search with ingredients taken out (HK_FILL_FAST bits; WRONG results): tools/fill_ablate.sh "<bench args>" v1 v2 ...
args=$1; shift
for v in "$@"; do
  HK_FILL_FAST=$v KSTATS_LINES=0 bash tools/kstats.sh abl_$v $args --steps 3 --warmup 1 --no-other-configs --no-power-probe > /dev/null 2>&1
  echo -n "HK_FILL_FAST=$v: "; python3 tools/kstat.py gpurun_out/abl_$v inpaint_fill
done
