#!/bin/bash
# tools/power_probe.sh "<bench args>" [lib]: package power / clocks sampled while bench.py runs its launches back to back
args="$1"; lib="$2"
rocm-smi --showmaxpower --showpower 2>/dev/null | grep -i "power\|GPU\[" | head -8
( HOMONIM_AMD_LIB=$lib python3 bench.py --steps 8000 --warmup 3 --no-cpu-baseline --no-nan-variant --no-parity $args > /tmp/pp_bench.txt 2>/dev/null ) &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk\|Power" | tr '\n' ' '; echo
  sleep 1
done
wait $pid
tail -1 /tmp/pp_bench.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('launch %.3f ms frac %.4f' % (r['avg_launch_ms'], r['frac']))"
