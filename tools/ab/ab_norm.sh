#!/bin/bash
# block statistics: the gain-blk-offset default model + configs[3] per library:  tools/ab/ab_norm.sh lib1 lib2 ...
for rep in 1 2; do for lib in "$@"; do for a in "--model gain-blk-offset" "--model gain-blk-offset --kernel 15 --bands 8" "--config 3 --no-end-to-end"; do HOMONIM_AMD_LIB=$lib python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $a 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-20s %-48s %8.3f ms/step frac %.4f parity=%s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed']))" "$(basename $lib)" "$a"; done; done; done
