"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / median / total."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]
    d[(n, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 0.002 * tot:
        continue
    print('%-50s grid %8s %5s %5s  n=%4d  med %9.1f us  total %9.1f us (%4.1f%%)' % (k + (len(v), sorted(v)[len(v) // 2], sum(v), 100 * sum(v) / tot)))
