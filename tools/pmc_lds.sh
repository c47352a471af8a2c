#!/bin/bash
# LDS / cache counters of bench.py's fused kernel (round 6: what binds the kernels wider than 15), one group per pass
# (separate runs, --kernel-trace only).  Usage (GPU box, repo root): tools/pmc_lds.sh <tag> [bench.py args ...]
# -> gpurun_out/pmcl_<tag>.txt (per-dispatch means of the dispatches of the kernel with the most time)
set -u
tag=$1; shift
export TMPDIR=/tmp
root=$(pwd)
groups=("SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE")
i=0
for g in "${groups[@]}"; do
  out=$root/gpurun_out/pmcl_${tag}_$i
  rm -rf "$out"
  (cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-trace -d "$out" -o run --output-format csv -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-nan-variant --no-other-configs --no-power-probe "$@" > "$out.log" 2>&1) || echo "pass $i ($g) failed: $(tail -2 $out.log)"
  i=$((i+1))
done
python3 - "$tag" "$@" <<'P' | tee "$root/gpurun_out/pmcl_$tag.txt"
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f'gpurun_out/pmcl_{tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fit_apply_kernel' in r['Kernel_Name'] or 'fit_list_kernel' in r['Kernel_Name'] or 'team' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[0][:90], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
names = sorted({k[0] for k in acc})
print('# bench.py', ' '.join(sys.argv[2:]))
for n in names:
    m = {c: sum(d.values()) / len(d) for (k, c), d in acc.items() if k == n}
    print(n)
    print('  ' + '  '.join(f'{c}={v:.4g}' for c, v in sorted(m.items())))
    w = m.get('SQ_WAVE_CYCLES')
    if w:
        print('  of wave cycles: active_any %.3f  wait_any %.3f  wait_inst_any %.3f  wait_inst_lds %.3f  valu %.3f  lds %.3f' % tuple(
            m.get(c, 0) / w for c in ('SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS')))
    if m.get('GRBM_GUI_ACTIVE'):
        cyc = m['GRBM_GUI_ACTIVE']   # GPU clocks of the launch (all 8 XCDs summed by the tool: / 8)
        if m.get('SQ_LDS_IDX_ACTIVE'):
            print('  LDS array busy: %.3f of the launch (SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8)); bank conflicts %.3f of the LDS cycles' % (
                m['SQ_LDS_IDX_ACTIVE'] / (256 * cyc / 8), m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']))
    if m.get('TCC_HIT_sum') is not None and m.get('TCC_MISS_sum'):
        print('  L2 hit rate %.3f' % (m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum'])))
    if m.get('FETCH_SIZE'):
        print('  fabric reads %.2f GB (FETCH_SIZE x 2, gfx950), writes %.2f GB' % (m['FETCH_SIZE'] * 2048 / 1e9, m.get('WRITE_SIZE', 0) * 1024 / 1e9))
P
