#!/bin/bash
# One-shot evidence run on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC passes of the
# headline and of the other configurations the round reports.
# Usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*; copy what is to be judged into profiles/)
set -u
tag=$1
export TMPDIR=/tmp
export HK_NO_FIRST_PROCESS_PROBE=1   # profiles of the bench process alone
root=$(pwd)
python3 bench.py > gpurun_out/${tag}_bench_full.json 2> gpurun_out/${tag}_bench_full.err
rm -rf gpurun_out/${tag}_stats
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/${tag}_stats" -o run --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline > "$root/gpurun_out/${tag}_stats.log" 2>&1)
python3 tools/kstat_by_grid.py gpurun_out/${tag}_stats/run_kernel_trace.csv 3 > gpurun_out/${tag}_kernel_stats_by_grid.csv
tools/pmc_collect.sh ${tag} > /dev/null 2>&1
tools/pmc_collect.sh ${tag}_nd2 --nodata 2 > /dev/null 2>&1
tools/pmc_collect.sh ${tag}_nd1 --nodata 1 > /dev/null 2>&1
tools/pmc_collect.sh ${tag}_gain --config 1 > /dev/null 2>&1
tools/pmc_collect.sh ${tag}_blk5 --model gain-blk-offset > /dev/null 2>&1
tools/pmc_collect.sh ${tag}_blk15 --model gain-blk-offset --kernel 15 --bands 8 > /dev/null 2>&1
tools/pmc_collect.sh ${tag}_params --params > /dev/null 2>&1
tail -1 gpurun_out/${tag}_bench_full.json
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs head -5
for v in "" _nd2 _nd1 _gain _blk5 _blk15 _params; do echo "== pmc ${tag}${v}"; python3 -c "
import json,sys
d=json.load(open('gpurun_out/pmc_${tag}${v}_summary.json'))
print({k:d.get(k) for k in ('valu_busy_fraction','cycles_per_valu_inst','valu_insts_per_wave','hbm_traffic_bytes','traffic_over_algorithmic')}, d['counters'].get('SQ_INSTS_VALU'), d['counters'].get('SQ_WAVES'))"; done
