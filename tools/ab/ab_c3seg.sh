run() { env $1 python3 bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end $2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-24s %-20s %8.3f ms/step frac %.4f parity=%s' % (sys.argv[1], sys.argv[2], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed']))" "$1" "$2"; }
for rep in 1 2; do for s in 0 64 128 512 1024; do run "A=1" "--seg-rows $s"; done; run "HK_BENCH_STREAMS=16" ""; run "HK_BENCH_STREAMS=12" "--seg-rows 512"; done
