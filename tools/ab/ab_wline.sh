#!/bin/bash
# A/B of the wide kernels' horizontal sums: the shipped library (LDS exchange lines) against _ab/lib_nowline.so (tools/mkvariant_full.sh nowline -DHK_WLINE=0: ds_bpermute)
run() { HOMONIM_AMD_LIB=$1 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nan-variant --no-other-configs --no-power-probe "${@:2}" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('%-18s %-40s %8.3f ms launch  %5.1f %%  parity=%s' % (sys.argv[1], ' '.join(sys.argv[2:]), r['avg_launch_ms'], 100 * r['frac'], d['parity_spot_check']['passed']))" "$(basename $1)" "${@:2}"; }
L1=homonim_amd/lib/libhomonim_hk.so; L2=_ab/lib_nowline.so
for args in "--kernel 17" "--kernel 21" "--kernel 31" "--kernel 41" "--kernel 63" "--kernel 31 --nodata 2" "--kernel 31 --nodata 1" "--model gain --kernel 31" "--model gain-blk-offset --kernel 31" "--kernel 31 --params" "--kernel 31 --no-thresh"; do
  for lib in $L1 $L2; do run $lib $args; done
done
