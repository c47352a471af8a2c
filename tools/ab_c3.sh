run() { env $1 python3 bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-60s %8.3f ms/step frac %.4f parity=%s' % (sys.argv[1], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed']))" "$1"; }
for e in "A=1" "HK_BENCH_C3_BANDS_PER_JOB=1" "HK_BENCH_C3_BANDS_PER_JOB=1 HK_BENCH_STREAMS=1" "HK_BENCH_C3_BANDS_PER_JOB=1 HK_BENCH_STREAMS=2" "HK_BENCH_C3_BANDS_PER_JOB=1 HK_BENCH_STREAMS=4" "HK_BENCH_C3_BANDS_PER_JOB=2 HK_BENCH_STREAMS=2" "HK_BENCH_C3_BANDS_PER_JOB=2 HK_BENCH_STREAMS=4" "HK_BENCH_C3_BANDS_PER_JOB=4 HK_BENCH_STREAMS=4" "HK_BENCH_STREAMS=4" "HK_BENCH_STREAMS=2" "HK_BENCH_STREAMS=1"; do run "$e"; done
