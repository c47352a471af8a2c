#!/bin/bash
# in-painting branch per library: tools/ab/ab_inpaint.sh lib1 lib2 ...
for rep in 1 2; do for lib in "$@"; do for a in "--nodata 3" "--nodata 4 --steps 4" "--nodata 3 --size 8192"; do HOMONIM_AMD_LIB=$lib python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $a 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-16s %-28s %8.3f ms/step frac %.4f parity=%s mism=%s fails=%s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed'], d['parity_spot_check']['bitwise_mismatches'], d['config']['r2_mask_failures_per_step']))" "$(basename $lib)" "$a"; done; done; done
