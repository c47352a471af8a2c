#!/bin/bash
# A/B of library builds on one box:  tools/ab/ab_libs.sh "<bench args>" lib1.so lib2.so ...   (each timed twice, interleaved)
args="$1"; shift
run() { HOMONIM_AMD_LIB=$1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-28s %-24s %8.3f ms launch  %8.3f ms/step parity=%s' % (sys.argv[1], sys.argv[2], r['avg_launch_ms'], d['ms_per_step'], d['parity_spot_check']['passed']))" "$(basename $1)" "$args"; }
for rep in 1 2; do for lib in "$@"; do run $lib; done; done
