#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per-dispatch averages."""
import collections
import csv
import glob
import sys


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k].add(r['Dispatch_Id'])
        for k, v in agg.items():
            if 'generate' in k or 'rocclr' in k:
                continue
            n = len(disp[k])
            print('%s  %s  dispatches=%d' % (d, k, n))
            for c, val in sorted(v.items()):
                print('    %-26s per-dispatch %.5g' % (c, val / n))
