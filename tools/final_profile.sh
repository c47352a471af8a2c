#!/bin/bash
# final evidence of the round: default bench line + rocprofv3 kernel stats of the same command
export TMPDIR=/tmp
root=$(pwd)
python3 bench.py > gpurun_out/r04f_bench_full.json 2> gpurun_out/r04f_bench_full.err; echo "bench rc=$?"
export HK_NO_FIRST_PROCESS_PROBE=1
rm -rf gpurun_out/r04f_stats
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$root/gpurun_out/r04f_stats" -o run --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline > "$root/gpurun_out/r04f_stats.log" 2>&1)
python3 tools/kstat_by_grid.py gpurun_out/r04f_stats/run_kernel_trace.csv 3 > gpurun_out/r04f_kernel_stats_by_grid.csv
head -3 gpurun_out/r04f_kernel_stats_by_grid.csv | cut -c1-200
