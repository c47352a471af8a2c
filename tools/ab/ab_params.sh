for i in 1 2 3 4 5 6 7 8; do python3 bench.py --params --steps 20 --warmup 3 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('params run: %8.3f ms launch  copy %6.0f GB/s parity=%s' % (r['avg_launch_ms'], r['copy_gbps_measured'], d['parity_spot_check']['passed']))"; done
