#!/bin/bash
# A/B of this tree against the round-5 tree (git worktree add _ab/r5 2d25abb; python -m homonim_amd.build there), interleaved, one box
run() { (cd $1 && python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe "${@:2}" 2>/dev/null | tail -1) | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-8s %-34s %8.3f ms launch %8.3f ms/step  %5.1f %%  parity=%s' % (sys.argv[1], ' '.join(sys.argv[2:]), r['avg_launch_ms'], d['ms_per_step'], 100 * r['frac'], d['parity_spot_check']['passed']))" "$(basename $(cd $1 && pwd))" "${@:2}"; }
for rep in 1 2 3; do for t in . _ab/r5; do run $t; done; done
for args in "--nodata 2" "--kernel 15" "--config 1" "--nodata 3 --steps 6" "--nodata 4 --steps 4" "--config 4 --steps 6 --no-end-to-end --no-projection"; do for t in . _ab/r5; do run $t $args; done; done
