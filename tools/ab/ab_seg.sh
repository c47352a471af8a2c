#!/bin/bash
# tools/ab/ab_seg.sh "<bench args>" seg1 seg2 ...
args="$1"; shift
run() { python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $args --seg-rows $1 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('seg %-4s %-36s %8.3f ms launch frac %.4f parity=%s' % (sys.argv[1], sys.argv[2], r['avg_launch_ms'], r['frac'], d['parity_spot_check']['passed']))" "$1" "$args"; }
for rep in 1 2; do for s in "$@"; do run $s; done; done
