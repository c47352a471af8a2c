"""Per (kernel, grid size) statistics from a rocprofv3 kernel-trace CSV: the default bench command launches one kernel
instantiation at several problem sizes (the headline's 16384^2 x 4 launch, the 4096^2 tiles of configs[4]), which the plain
per-kernel stats average together.  Usage: python tools/kstat_by_grid.py <run_kernel_trace.csv> [min_calls] > summary.csv"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (r['Kernel_Name'], int(r['Grid_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']), int(r['Workgroup_Size_X']),
           int(r['VGPR_Count']), int(r['LDS_Block_Size']))
    rows[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = csv.writer(sys.stdout)
w.writerow(['Name', 'Grid_X', 'Grid_Y', 'Grid_Z', 'Workgroup_X', 'VGPRs', 'LDS_bytes', 'Calls', 'TotalNs', 'AverageNs', 'MedianNs', 'MinNs', 'MaxNs'])
for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    if len(d) < min_calls:
        continue
    d.sort()
    w.writerow([key[0][:110], *key[1:], len(d), sum(d), round(sum(d) / len(d), 1), d[len(d) // 2], d[0], d[-1]])
