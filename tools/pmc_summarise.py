"""Per-launch means of the PMC counters collected by tools/pmc_collect.sh for hk::fit_apply_kernel dispatches."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
acc = {}
for path in glob.glob(f'gpurun_out/pmc_{tag}_*/**/*counter_collection.csv', recursive=True):
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if os.environ.get('HK_PMC_KERNEL', 'fit_apply_kernel') not in row['Kernel_Name']:
                continue
            per.setdefault(row['Counter_Name'], {}).setdefault(row['Dispatch_Id'], 0.0)
            per[row['Counter_Name']][row['Dispatch_Id']] += float(row['Counter_Value'])
    for name, d in per.items():
        vals = list(d.values())
        acc[name] = sum(vals) / len(vals)
out = {'counters': acc}
c = acc
if 'FETCH_SIZE' in c:
    out['hbm_read_bytes'] = c['FETCH_SIZE'] * 1024 * 2  # gfx950 correction (MI355X_MICROARCH.md)
if 'WRITE_SIZE' in c:
    out['hbm_write_bytes'] = c['WRITE_SIZE'] * 1024
if 'SQ_ACTIVE_INST_VALU' in c and 'GRBM_GUI_ACTIVE' in c:
    out['valu_busy_fraction'] = c['SQ_ACTIVE_INST_VALU'] * 4 / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)
    out['cycles_per_valu_inst'] = c['SQ_ACTIVE_INST_VALU'] * 4 / c['SQ_INSTS_VALU']
    out['valu_insts_per_wave'] = c['SQ_INSTS_VALU'] / c['SQ_WAVES']
print(json.dumps(out, indent=1))
