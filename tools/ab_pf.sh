run() { HK_LDS_PAD=$2 HOMONIM_AMD_LIB=$1 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe $3 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-14s pad %-6s %-22s %8.3f ms launch frac %.4f parity=%s' % (sys.argv[1], sys.argv[2], sys.argv[3], r['avg_launch_ms'], r['frac'], d['parity_spot_check']['passed']))" "$(basename $1)" "$2" "$3"; }
for a in "--model gain" "--config 1"; do for rep in 1 2; do
for l in wpb4 pf1 pf3 pf4; do run _ab/lib_$l.so 4096 "$a"; done; run _ab/lib_pf4.so 6144 "$a"; run _ab/lib_pf3.so 6144 "$a"
done; done
