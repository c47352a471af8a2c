run() { HOMONIM_AMD_LIB=$1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-nan-variant $2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-28s %-24s %8.3f ms launch  %8.3f ms/step frac %.4f parity=%s' % (sys.argv[1], sys.argv[2], r['avg_launch_ms'], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed']))" "$(basename $1)" "$2"; }
# usage: tools/ab/ab_quick.sh "<args1>|<args2>|..." lib1 lib2 ...   (each configuration on each library, twice)
IFS='|' read -ra CFGS <<< "$1"; shift
for cfg in "${CFGS[@]}"; do for rep in 1 2; do for lib in "$@"; do run $lib "$cfg"; done; done; done
