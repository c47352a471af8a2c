#!/bin/bash
# ab_env.sh "<bench args>" "ENV1=.. ENV2=.." "ENV=.." ...
args="$1"; shift
run() { env $1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-34s %-26s %8.3f ms launch  %8.3f ms/step parity=%s' % (sys.argv[1], sys.argv[2], r['avg_launch_ms'], d['ms_per_step'], d['parity_spot_check']['passed']))" "$1" "$args"; }
for rep in 1 2; do for e in "$@"; do run "$e"; done; done
