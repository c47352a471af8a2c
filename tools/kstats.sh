#!/bin/bash
# tools/kstats.sh TAG <bench.py args...>: rocprofv3 per-kernel statistics of one bench.py configuration -> gpurun_out/TAG/
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
export TMPDIR=/tmp
export HK_NO_FIRST_PROCESS_PROBE=1   # profiles of the bench process alone
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/$tag" -o run --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-parity --no-nan-variant "$@" > "$root/gpurun_out/$tag.log" 2>&1)
cut -c1-170 "$root/gpurun_out/$tag/run_kernel_stats.csv" | head -${KSTATS_LINES:-9}
