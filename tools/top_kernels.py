"""Print the top rows of a rocprofv3 *_kernel_stats.csv.   python tools/top_kernels.py FILE [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print('%-100s %7s %12s %10s %6s' % ('kernel', 'calls', 'total us', 'avg us', '%'))
for r in rows[:n]:
    print('%-100s %7s %12.1f %10.2f %6.2f' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                             float(r['AverageNs']) / 1e3, float(r['Percentage'])))
