#!/bin/bash
# The measurement table of DESIGN.md section 6 from one box:  tools/sweep_table.sh > gpurun_out/sweep.txt
run() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-46s %8.3f ms  %9.0f Mpx*b/s  %6.0f GB/s  %5.1f %%  parity=%s' % (' '.join(sys.argv[1:]) or '(headline)', r['avg_launch_ms'], d['value'], r['achieved'], 100 * r['frac'], d['parity_spot_check']['passed']))" "$@"; }
run
run
run --nodata 2
run --nodata 1
run --kernel 3
run --no-thresh
run --model gain
run --model gain --size 8192
run --kernel 7
run --kernel 9
run --kernel 15
run --kernel 15 --nodata 1
run --nodata 6
run --kernel 15 --nodata 6
run --model gain-blk-offset
run --model gain-blk-offset --kernel 15 --bands 8
run --nodata 3
run --params
run --model gain --params
run --nodata 4 --steps 3
run --model gain-blk-offset --nodata 2
run --model gain --kernel 7
run --model gain --kernel 11
run --model gain --kernel 15
run --model gain-blk-offset --kernel 9
# kernels wider than 15 (round 5: hsum_wide builds)
run --kernel 17
run --kernel 21
run --kernel 31
run --kernel 17 --nodata 2
run --kernel 21 --nodata 2
run --kernel 31 --nodata 2
run --kernel 31 --nodata 1
run --model gain --kernel 31
run --model gain-blk-offset --kernel 31
