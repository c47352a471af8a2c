#!/bin/bash
# in-painting step per environment setting: tools/ab_fill_env.sh "ENV=.." "ENV=.." ...   (bench.py --nodata 3 / 4, twice each)
for rep in 1 2; do for e in "$@"; do for a in "--nodata 3" "--nodata 4 --steps 4"; do env $e python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $a 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-28s %-24s %8.3f ms/step parity=%s mism=%s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['parity_spot_check']['passed'], d['parity_spot_check']['bitwise_mismatches']))" "$e" "$a"; done; done; done
