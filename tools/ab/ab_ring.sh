#!/bin/bash
# tools/ab/ab_ring.sh lib "<bench args>" mode1 mode2 ...: HK_USE_RING modes of one library, twice each
lib=$1; args="$2"; shift; shift
run() { HK_USE_RING=$1 HOMONIM_AMD_LIB=$lib python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe --no-end-to-end $args 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('ring %-3s %-44s %8.3f ms/step frac %.4f parity=%s mism=%s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed'], d['parity_spot_check']['bitwise_mismatches']))" "$1" "$args"; }
for rep in 1 2; do for m in "$@"; do run $m; done; done
