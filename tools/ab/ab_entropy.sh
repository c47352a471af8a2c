for nd in 5 0 5 0; do python3 bench.py --nodata $nd --no-cpu-baseline --no-nan-variant --no-other-configs 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['power']
print('nodata %s: %.3f ms launch frac %.4f  %s W %s MHz parity=%s fails=%s' % (sys.argv[1], d['roofline']['avg_launch_ms'], d['roofline']['frac'], p['package_watts'], p['sclk_mhz'], d['parity_spot_check']['passed'], d['config']['r2_mask_failures_per_step']))" $nd; done
