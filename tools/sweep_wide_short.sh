#!/bin/bash
# A short form of sweep_wide.sh (one box calibration line + the wide kernels the round-6 verdict names):  tools/sweep_wide_short.sh > gpurun_out/x.txt
run() { python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-46s %8.3f ms  %9.0f Mpx*b/s  %6.0f GB/s  %5.1f %%  parity=%s' % (' '.join(sys.argv[1:]) or '(headline)', r['avg_launch_ms'], d['value'], r['achieved'], 100 * r['frac'], d['parity_spot_check']['passed']))" "$@"; }
run
for k in 15 17 21 31 41 63; do run --kernel $k; done
for k in 15 31 63; do run --kernel $k --nodata 2; done
run --kernel 31 --nodata 1
for m in gain gain-blk-offset; do run --model $m --kernel 31; done
run --kernel 31 --params
