"""Print the average duration (ms) of kernels whose name contains a pattern, from a rocprofv3 kernel_stats csv dir."""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
for path in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if pat in row['Name']:
            print(f"{row['Name'][:60]}: calls {row['Calls']} avg {float(row['AverageNs'])/1e6:.4f} ms min {float(row['MinNs'])/1e6:.4f}")
