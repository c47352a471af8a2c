run() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-40s %8.3f ms launch  frac %.4f parity=%s' % (' '.join(sys.argv[1:]), r['avg_launch_ms'], r['frac'], d['parity_spot_check']['passed']))" "$@"; }
export HOMONIM_AMD_LIB=_ab/lib_pf1.so
for k in 17 25 31 39; do echo "ring default:"; run --kernel $k; echo "ring 0:"; HK_USE_RING=0 run --kernel $k; done
