#!/bin/bash
# One-shot evidence run on the GPU box: default bench line, rocprofv3 kernel stats of the same command, PMC passes.
# Usage: tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>_*)
set -u
tag=$1
export TMPDIR=/tmp
root=$(pwd)
python3 bench.py > gpurun_out/${tag}_bench_full.json 2> gpurun_out/${tag}_bench_full.err
rm -rf gpurun_out/${tag}_stats
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/${tag}_stats" -o run --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline > "$root/gpurun_out/${tag}_stats.log" 2>&1)
tools/pmc_collect.sh ${tag} > /dev/null 2>&1
tail -1 gpurun_out/${tag}_bench_full.json
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs head -5
cat gpurun_out/pmc_${tag}_summary.json
