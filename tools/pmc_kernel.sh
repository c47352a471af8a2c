#!/bin/bash
# tools/pmc_kernel.sh <tag> <kernel name substring> <bench args...>: SQ counters of ONE kernel of a bench.py run (two --pmc passes,
# --kernel-trace only), per-dispatch means -> stdout
tag=$1; pat=$2; shift; shift
export TMPDIR=/tmp
root=$(pwd)
i=0
for g in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  out=$root/gpurun_out/pmck_${tag}_$i; rm -rf "$out"
  (cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-trace -d "$out" -o run --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-nan-variant --no-other-configs --no-power-probe "$@" > "$out.log" 2>&1)
  i=$((i+1))
done
python3 - "$tag" "$pat" <<'P'
import csv, glob, sys, collections
tag, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f'gpurun_out/pmck_{tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print({k: round(v) for k, v in m.items()}, 'dispatches', {k: len(v) for k, v in acc.items()}.get('SQ_WAVES'))
if 'SQ_WAVES' in m:
    w = m['SQ_WAVES']
    print('per wave: VALU %.0f  SALU %.0f  LDS %.0f  VMEM_RD %.0f;  VALU active / wave cycles %.3f, wait_any %.3f, wait_inst %.3f, active_any %.3f' % (
        m.get('SQ_INSTS_VALU', 0) / w, m.get('SQ_INSTS_SALU', 0) / w, m.get('SQ_INSTS_LDS', 0) / w, m.get('SQ_INSTS_VMEM_RD', 0) / w,
        m.get('SQ_ACTIVE_INST_VALU', 0) / m['SQ_WAVE_CYCLES'], m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES'],
        m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES'], m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']))
P
