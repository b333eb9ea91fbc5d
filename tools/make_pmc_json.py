#!/usr/bin/env python3
"""profiles/pmc_traffic.json from rocprofv3 --pmc runs of `python3 bench.py --steps 2 --warmup 1 --cpu-sample 0`
(one directory per counter set, CSV output).  Usage: make_pmc_json.py PREFIX   (directories PREFIX_*)

HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes x 1024): FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950.  VALU issue cycles per launch =
4 x (FMA_F64 + ADD_F64 + MUL_F64 + CVT) + 2 x (the other vector instructions): fp64-rate instructions occupy a SIMD
for 4 cycles per wave64, 32-bit ones for 2 (tools/microbench/valu_rate.hip)."""
import collections
import csv
import glob
import json
import sys


def collect(prefix):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(prefix + '_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k][r['Counter_Name']].add(r['Dispatch_Id'])
    return {k: {c: v / len(disp[k][c]) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    per = collect(sys.argv[1])
    out = {'note': __doc__.split('\n\n', 1)[1].replace('\n', ' '), 'units_per_launch': 1048576}
    for key, kern in (('analysis', 'k_analysis_fast<true>'), ('pack', 'k_pack'), ('allocate', 'k_alloc_first')):
        c = per.get(kern)
        if not c:
            continue
        e = {'kernel': kern}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            e.update(fetch_size_kb=c['FETCH_SIZE'], write_size_kb=c['WRITE_SIZE'],
                     hbm_bytes_per_launch=(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024)
        if 'SQ_INSTS_VALU' in c:
            f64 = sum(c.get(n, 0.0) for n in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_CVT'))
            e.update(valu_insts_per_launch=c['SQ_INSTS_VALU'], valu_fp64_rate_insts_per_launch=f64,
                     valu_issue_cycles_per_launch=4 * f64 + 2 * (c['SQ_INSTS_VALU'] - f64),
                     lds_insts_per_launch=c.get('SQ_INSTS_LDS'), lds_active_cycles_per_launch=c.get('SQ_LDS_IDX_ACTIVE'),
                     lds_bank_conflict_cycles_per_launch=c.get('SQ_LDS_BANK_CONFLICT'))
        out[key] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
