#!/bin/bash
# tools/ab/ab_once.sh "<args1>|<args2>|..." lib1 lib2 ...: each configuration on each library ONCE, 8 steps (screening; confirm with ab_quick.sh)
run() { HOMONIM_AMD_LIB=$1 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-22s %-40s %8.3f ms launch  frac %.4f parity=%s' % (sys.argv[1], sys.argv[2], r['avg_launch_ms'], r['frac'], d['parity_spot_check']['passed']))" "$(basename $1)" "$2"; }
IFS='|' read -ra CFGS <<< "$1"; shift
for cfg in "${CFGS[@]}"; do for lib in "$@"; do run $lib "$cfg"; done; done
