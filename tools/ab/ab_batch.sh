#!/bin/bash
# tools/ab/ab_batch.sh [config ...]: configs[3] / configs[4] with their jobs in 0 (one launch per job) / 1 / 2 / 4 batched launches
run() { env $2 python3 bench.py $1 --no-cpu-baseline --no-end-to-end --no-other-configs --no-power-probe 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-18s %-48s %8.3f ms/step frac %.4f parity=%s' % (sys.argv[2], sys.argv[1], d['ms_per_step'], r['frac'], d['parity_spot_check']['passed']))" "$1" "$2"; }
cfgs=${@:-3 4}
for rep in 1 2; do
for c in $cfgs; do
for b in 0 1 2 4; do run "--config $c --steps 8 --warmup 2 --batches $b" "A=1"; done
run "--config $c --steps 8 --warmup 2 --batches 1" "HK_BATCH_SEGS=0"
run "--config $c --steps 8 --warmup 2 --batches 2" "HK_BATCH_SEGS=0"
done; done
