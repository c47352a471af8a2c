#!/bin/bash
# in-painting step against HK_FILL_CONT (lanes that must still be open for the packed search to go on): tools/ab_fill_cont.sh v1 v2 ...
for c in "$@"; do for a in "--nodata 3" "--nodata 4 --steps 4"; do HK_FILL_CONT=$c python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $a 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('cont %-4s %-24s %8.3f ms/step parity=%s mism=%s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['parity_spot_check']['passed'], d['parity_spot_check']['bitwise_mismatches']))" "$c" "$a"; done; done
