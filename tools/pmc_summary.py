"""Average rocprofv3 --pmc counter values per kernel:  python tools/pmc_summary.py <dir> [kernel substring]"""
import collections
import csv
import glob
import sys

sub = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
        if sub in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-32s n=%3d  mean %.4g' % (c, len(v), sum(v) / len(v)))
