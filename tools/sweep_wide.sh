#!/bin/bash
# Kernels wider than 15 on the headline workload:  tools/sweep_wide.sh > gpurun_out/sweep_wide.txt
run() { python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-46s %8.3f ms  %9.0f Mpx*b/s  %6.0f GB/s  %5.1f %%  parity=%s' % (' '.join(sys.argv[1:]) or '(headline)', r['avg_launch_ms'], d['value'], r['achieved'], 100 * r['frac'], d['parity_spot_check']['passed']))" "$@"; }
for k in 15 17 19 21 23 25 27 29 31 33 35 37 39 41 47 63; do run --kernel $k; done
for k in 15 17 19 21 23 27 31 33 35 37 39; do run --kernel $k --nodata 2; done
for k in 17 21 31; do run --kernel $k --nodata 1; done
for k in 21 31; do run --kernel $k --nodata 6; done
for m in gain gain-blk-offset; do for k in 15 17 31 35; do run --model $m --kernel $k; done; done
for k in 11 13 15; do run --model gain --kernel $k --nodata 2; done
for k in 31 35; do run --no-thresh --kernel $k; done
