#!/bin/bash
# tools/power_probe2.sh "<bench args>" steps: package power / clock while the configuration repeats (~10 s)
args="$1"; steps=${2:-1000}
( python3 bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-nan-variant --no-parity --no-other-configs --no-power-probe --no-end-to-end $args > /tmp/pp_bench.txt 2>/dev/null ) &
pid=$!
sleep 9
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | awk '/Package Power/ {p=$NF} /sclk clock level/ {gsub(/[()Mhz]/,"",$NF); s=$NF} END {printf "%s W %s MHz | ", p, s}'
  sleep 0.7
done
echo
wait $pid
tail -1 /tmp/pp_bench.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%s: %.3f ms/step frac %.4f' % (sys.argv[1], d['ms_per_step'], r['frac']))" "$args"
