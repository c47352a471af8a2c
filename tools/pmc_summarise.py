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
# the bench.py arguments the passes ran with (tools/pmc_collect.sh hands them on): bench.py reports `roofline.traffic`
# from this file only at exactly this configuration
import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--config', type=int, default=2)
ap.add_argument('--size', type=int, default=None), ap.add_argument('--bands', type=int, default=None)
ap.add_argument('--model', default=None), ap.add_argument('--kernel', type=int, default=None)
ap.add_argument('--nodata', type=int, default=0), ap.add_argument('--no-thresh', action='store_true')
ap.add_argument('--params', action='store_true')
cfg, _ = ap.parse_known_args(sys.argv[2:])
presets = {1: dict(model='gain', kernel=5, size=8192, bands=4), 2: dict(model='gain-offset', kernel=5, size=16384, bands=4),
           3: dict(model='gain-blk-offset', kernel=15, size=16384, bands=8), 4: dict(model='gain-offset', kernel=5, size=4096, bands=4)}
for k_, v_ in presets[cfg.config].items():  # the same presets as bench.py --config
    if getattr(cfg, k_) is None:
        setattr(cfg, k_, v_)
out['config'] = dict(model=cfg.model, kernel=cfg.kernel, size=cfg.size, bands=cfg.bands, nodata=cfg.nodata,
                     no_thresh=cfg.no_thresh, params=cfg.params)
if 'hbm_read_bytes' in out and 'hbm_write_bytes' in out:
    out['hbm_traffic_bytes'] = out['hbm_read_bytes'] + out['hbm_write_bytes']
    out['algorithmic_bytes'] = 12 * cfg.size * cfg.size * cfg.bands
    out['traffic_over_algorithmic'] = out['hbm_traffic_bytes'] / out['algorithmic_bytes']
out['note'] = ('rocprofv3 --pmc passes (tools/pmc_collect.sh: separate runs, --kernel-trace only) of bench.py at this config; '
               'per-launch means over the fit_apply_kernel dispatches; FETCH_SIZE (KiB) doubled per the gfx950 correction of '
               'MI355X_MICROARCH.md; SQ_* are quad-cycles / wave-instructions')
print(json.dumps(out, indent=1))
