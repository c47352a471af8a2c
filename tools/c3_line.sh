c3() { python3 bench.py --config 3 --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-projection "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-60s %8.3f ms/step  frac %.4f parity=%s' % (sys.argv[1], d['ms_per_step'], d['roofline']['frac'], d['parity_spot_check']['passed']))" "$LABEL $*"; }
