#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the --pmc passes of tools/profile.sh.   Usage: make_pmc_json.py gpurun_out/prof_TAG

HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes x 1024): FETCH_SIZE is doubled as MI355X_MICROARCH.md
prescribes for 16-byte-per-lane streaming reads on gfx950.  Every wave64 vector instruction occupies its SIMD for one
quad-cycle (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU on every kernel here, packed binary32 included), so VALU issue cycles
per launch = 4 x SQ_INSTS_VALU.  `source_sha` identifies the kernel sources (bench.py source_sha()): bench.py only
quotes `traffic` from this file when it runs the very sources the counters were collected from."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def collect(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in glob.glob(d + '/pmc*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            # list-mode launches of the redo pass are tiny: keep them apart from the full-batch launches of the same kernel
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            disp[k][r['Counter_Name']].add(r['Dispatch_Id'])
    return {k: {c: v / len(disp[k][c]) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    from bench import source_sha
    per = collect(sys.argv[1])
    # sound units one full-batch launch covers: the profiled bench line says how many frames and launches a step had
    units = 1048576
    try:
        b = json.loads(open(os.path.join(sys.argv[1], 'bench_under_rocprof.json')).read().strip().splitlines()[-1])
        units = int(b['roofline']['stereo_frames_per_launch'] * b['config']['channels'])
    except Exception:   # noqa: BLE001
        pass
    out = {'note': __doc__.split('\n\n', 1)[1].replace('\n', ' '), 'units_per_launch': units, 'source_sha': source_sha()}
    for key, kern in (('analysis', 'k_analysis_spec<false>'), ('pack', 'k_pack<true, true>'), ('allocate', 'k_alloc_first'),
                      ('redo', 'k_analysis_fast<true>')):
        c = per.get(kern)
        if not c:
            continue
        e = {'kernel': kern}
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            e.update(fetch_size_kb=c['FETCH_SIZE'], write_size_kb=c['WRITE_SIZE'],
                     hbm_bytes_per_launch=(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024)
        if 'SQ_INSTS_VALU' in c:
            e.update(valu_insts_per_launch=c['SQ_INSTS_VALU'], valu_issue_cycles_per_launch=4 * c['SQ_INSTS_VALU'],
                     valu_active_quadcycles_per_launch=c.get('SQ_ACTIVE_INST_VALU'),
                     lds_insts_per_launch=c.get('SQ_INSTS_LDS'), lds_active_cycles_per_launch=c.get('SQ_LDS_IDX_ACTIVE'),
                     lds_bank_conflict_cycles_per_launch=c.get('SQ_LDS_BANK_CONFLICT'), busy_cu_cycles_per_launch=c.get('SQ_BUSY_CU_CYCLES'))
        if kern in ('k_alloc_first', 'k_analysis_fast<true>'):
            e['note'] = 'average over the full-batch launches AND the small list-mode launches of the redo pass'
        out[key] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
