#!/bin/bash
# tools/power_phases.sh [secs-per-phase]: tools/ubench_power beside a rocm-smi sampler -> per phase: package power, sclk, rate
secs=${1:-4}
out=${2:-gpurun_out/r03_power_phases}
mkdir -p gpurun_out
( while true; do
    t=$(date +%s.%N)
    rocm-smi --showpower --showclocks 2>/dev/null | awk -v t=$t '/Package Power/ {p=$NF} /sclk clock level/ {gsub(/[()Mhz]/,"",$NF); s=$NF} END {print t, p, s}'
  done ) > ${out}_samples.txt &
sampler=$!
tools/ubench_power $secs > ${out}_phases.txt
kill $sampler
python3 - ${out}_phases.txt ${out}_samples.txt <<'PY'
import sys, statistics
phases = [l.rstrip('\n').split('\t') for l in open(sys.argv[1]) if l.startswith('PHASE')]
samples = []
for l in open(sys.argv[2]):
    p = l.split()
    if len(p) == 3:
        try: samples.append((float(p[0]), float(p[1]), float(p[2])))
        except ValueError: pass
print('%-78s %8s %8s %12s' % ('phase', 'watts', 'sclk', 'rate'))
for _, name, t0, t1, launches, rate, unit in phases:
    t0, t1 = float(t0), float(t1)
    sel = [(w, c) for (t, w, c) in samples if t0 + 1.0 <= t <= t1 - 0.3]
    if sel:
        w = statistics.median(x[0] for x in sel); c = statistics.median(x[1] for x in sel)
        print('%-78s %8.0f %8.0f %12s %s  (%d samples)' % (name, w, c, rate, unit, len(sel)))
    else:
        print('%-78s       --       -- %12s %s' % (name, rate, unit))
PY
