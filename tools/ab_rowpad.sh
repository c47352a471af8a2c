#!/bin/bash
# row stride experiments: tools/ab_rowpad.sh "<bench args>" pad1 pad2 ...
args="$1"; shift
run() { HK_BENCH_ROW_PAD=$1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $args 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('pad %-6s %-28s %8.3f ms launch frac %.4f copy %6.0f GB/s parity=%s' % (sys.argv[1], sys.argv[2], r['avg_launch_ms'], r['frac'], r['copy_gbps_measured'] or 0, d['parity_spot_check']['passed']))" "$1" "$args"; }
for rep in 1 2; do for p in "$@"; do run $p; done; done
