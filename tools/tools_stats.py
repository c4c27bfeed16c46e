import csv, glob, sys
tag = sys.argv[1]
f = sorted(glob.glob('gpurun_out/prof_%s/*/*_kernel_stats.csv' % tag))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in rows[:n]:
    print("%-78s calls=%5s avg=%8.1fus tot=%7.2fms %5.1f%%" % (r['Name'][:78], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
print('total ms', tot / 1e6, 'launches', sum(int(r['Calls']) for r in rows), f)
