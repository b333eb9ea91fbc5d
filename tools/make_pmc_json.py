#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the passes of tools/profile.sh.
Usage: make_pmc_json.py gpurun_out/prof_CONFIG2 [gpurun_out/prof_CONFIG3]   (second: the same passes with --signal pink --modes detect)

HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes x 1024): FETCH_SIZE is doubled as MI355X_MICROARCH.md
prescribes for 16-byte-per-lane streaming reads on gfx950.  Every wave64 vector instruction occupies its SIMD for one
quad-cycle (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU on every kernel here, packed binary32 included), so VALU issue cycles
per launch = 4 x SQ_INSTS_VALU, to be held against 4 SIMDs x SQ_BUSY_CU_CYCLES.  `profiled_launch_ms` is the kernel's
average full-size launch in the --kernel-trace pass of the same command.  `source_sha` identifies the kernel sources
(bench.py source_sha()): bench.py only quotes `traffic` and `roofline_valu` from this file when it runs the very
sources the counters were collected from."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def collect(d):
    """per kernel: counter -> average over its FULL-SIZE dispatches (the list-mode launches of the redo pass are tiny:
    a dispatch counts when its value is at least a quarter of the kernel's largest for that counter)"""
    vals = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(d + '/pmc*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            vals[short(r['Kernel_Name'])][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    out = {}
    for k, cs in vals.items():
        out[k] = {}
        for c, per in cs.items():
            big = [v for v in per.values() if v >= 0.25 * max(per.values())]
            out[k][c] = sum(big) / len(big)
    return out


def launch_ms(d):
    per = collections.defaultdict(list)
    for f in glob.glob(d + '/trace/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            per[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    return {k: (lambda big: sum(big) / len(big))([x for x in v if x >= 0.25 * max(v)]) for k, v in per.items()}


def section(d, kernels):
    per, ms = collect(d), launch_ms(d)
    units = 2097152
    try:
        b = json.loads(open(os.path.join(d, 'bench_under_rocprof.json')).read().strip().splitlines()[-1])
        units = int(b['config']['frames_per_gpu'] * b['config']['channels'])
    except Exception:   # noqa: BLE001
        pass
    out = {'units_per_launch': units}
    for key, kern in kernels:
        c = per.get(kern)
        if not c:
            continue
        e = {'kernel': kern, 'profiled_launch_ms': ms.get(kern)}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            e.update(fetch_size_kb=c['FETCH_SIZE'], write_size_kb=c['WRITE_SIZE'],
                     hbm_bytes_per_launch=(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024)
        if 'SQ_INSTS_VALU' in c:
            e.update(valu_insts_per_launch=c['SQ_INSTS_VALU'], valu_issue_cycles_per_launch=4 * c['SQ_INSTS_VALU'],
                     valu_active_quadcycles_per_launch=c.get('SQ_ACTIVE_INST_VALU'),
                     lds_insts_per_launch=c.get('SQ_INSTS_LDS'), lds_active_cycles_per_launch=c.get('SQ_LDS_IDX_ACTIVE'),
                     lds_bank_conflict_cycles_per_launch=c.get('SQ_LDS_BANK_CONFLICT'), busy_cu_cycles_per_launch=c.get('SQ_BUSY_CU_CYCLES'))
        out[key] = e
    return out


def main():
    from bench import source_sha
    out = {'note': __doc__.split('\n\n', 1)[1].replace('\n', ' '), 'source_sha': source_sha(), 'sections': {}}
    out['sections']['config2'] = section(sys.argv[1], (('analysis', 'k_analysis_spec<false>'), ('pack', 'k_pack<true, true>'),
                                                       ('allocate', 'k_alloc_first'), ('redo', 'k_analysis_fast<true>')))
    if len(sys.argv) > 2:
        out['sections']['config3'] = section(sys.argv[2], (('analysis', 'k_detect_features<true>'), ('mdct_long', 'k_mdct_bands<true>'),
                                                           ('mdct_mixed', 'k_mdct_bands<false>'), ('pack', 'k_pack<false, true>'),
                                                           ('allocate', 'k_alloc_first')))
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
