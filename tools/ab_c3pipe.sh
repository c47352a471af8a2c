#!/bin/bash
# tools/ab_c3pipe.sh: configs[3] with the statistics of batch g + 1 beside the fit of batch g (two streams, event-ordered)
run() { env $2 python3 bench.py --config 3 --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end $1 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-52s %-16s %8.3f ms/step frac %.4f parity=%s' % (sys.argv[2], sys.argv[1], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed']))" "$1" "$2"; }
for rep in 1 2; do
run "--batches 4" "A=1"
for b in 2 4 8 16; do
run "--batches $b" "HK_BENCH_C3_PIPE=1"
run "--batches $b" "HK_BENCH_C3_PIPE=1 HK_HIGH_PRIO_STREAMS=1"
done; done
