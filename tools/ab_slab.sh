#!/bin/bash
# one slab vs one allocation per plane, 6 processes each, interleaved:  tools/ab_slab.sh "<bench args>"
for rep in 1 2 3 4 5 6; do for slab in 1 0; do echo -n "one_slab=$slab: "; HK_BENCH_ONE_SLAB=$slab python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity $1 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'])"; done; done
