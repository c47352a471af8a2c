#!/bin/bash
run() { python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read())
r = d['roofline']
print('CERT=%s %-30s %8.3f ms launch  %8.3f ms/step parity=%s' % (os.environ.get('HK_CERT_ONLY','-'), ' '.join(sys.argv[1:]) or '(headline)', r['avg_launch_ms'], d['ms_per_step'], d['parity_spot_check']['passed']))" "$@"; }
for args in "" "--nodata 1" "--nodata 2" "--kernel 3" "--kernel 7" "--nodata 3 --size 8192" "--nodata 4 --size 8192"; do
  for c in 0 1 0 1; do HK_CERT_ONLY=$c run $args; done
done
