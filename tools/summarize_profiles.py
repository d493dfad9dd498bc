"""Turns the rocprofv3 outputs of one round (gpurun_out/<dir>/*.csv) into the committed summaries under profiles/.

  python tools/summarize_profiles.py <tag> <trace_dir> [<pmc_sq_dir> <pmc_fetch_dir> <pmc_write_dir>] [--steps N]

* <trace_dir>: `rocprofv3 --kernel-trace --stats --output-format csv` of `python3 bench.py ...`   -> profiles/<tag>_kernel_stats.csv
* PMC dirs (separate passes, as MI355X_MICROARCH.md prescribes): SQ_* + GRBM_GUI_ACTIVE, FETCH_SIZE, WRITE_SIZE
  -> profiles/<tag>_pmc_summary.json, per kernel: MFMA busy fraction, LDS activity, HBM-side bytes per launch
  (FETCH_SIZE is doubled: gfx950 tallies 128-byte requests at 64 bytes; WRITE_SIZE is exact; both in KB).
bench.py reads profiles/latest_pmc_summary.json (a copy of the newest summary) for `roofline.traffic`.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(DF16b|DF16_|f)?', name)
    if m:
        return m.group(1) + {'DF16b': '<bf16>', 'DF16_': '<f16>', 'f': '<f32>', None: ''}[m.group(2)]
    m = re.match(r'(?:void )?([a-z_0-9]+)(<[^(]*)?\(', name)
    return m.group(1) if m else name[:60]


def counters(path, wanted):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    f = glob.glob(os.path.join(path, '*counter_collection.csv'))
    if not f:
        return d
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] in wanted:
            d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return d


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opts = dict(a[2:].split('=', 1) for a in sys.argv[1:] if a.startswith('--') and '=' in a)      # --workload=wrn-28-10 --dtype=fp16
    tag, trace = args[0], args[1]
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    stats = glob.glob(os.path.join(trace, '*kernel_stats.csv'))[0]
    shutil.copy(stats, os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv'))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(stats)):
        k = short(r['Name'])
        dur[k][0] += int(r['Calls'])
        dur[k][1] += float(r['TotalDurationNs'])
    out = {'tag': tag, 'workload': opts.get('workload', 'wrn-28-10'), 'dtype': opts.get('dtype', 'fp16'), 'kernels': {}}
    if len(args) >= 5:
        sq = counters(args[2], {'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT',
                                'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'GRBM_GUI_ACTIVE'})
        fe = counters(args[3], {'FETCH_SIZE'})
        wr = counters(args[4], {'WRITE_SIZE'})
        for k in sorted(dur, key=lambda k: -dur[k][1]):
            calls, total = dur[k]
            e = {'launches_in_trace': calls, 'avg_us': round(total / calls / 1e3, 2)}
            if k in sq and sq[k].get('GRBM_GUI_ACTIVE'):
                m = {c: sum(v) / len(v) for c, v in sq[k].items()}
                cyc = m['GRBM_GUI_ACTIVE'] / 8.0                      # counter is summed over the 8 XCDs
                e['mfma_busy_frac'] = round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024 * cyc), 4)      # 256 CUs x 4 SIMDs
                e['lds_active_frac'] = round(m.get('SQ_LDS_IDX_ACTIVE', 0.0) / (256 * cyc), 4)
                e['lds_conflict_frac_of_active'] = round(m.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(m.get('SQ_LDS_IDX_ACTIVE', 0.0), 1.0), 4)
                wc = max(m.get('SQ_WAVE_CYCLES', 0.0), 1.0)
                e['wave_cycles_split'] = {'waiting': round(m.get('SQ_WAIT_ANY', 0.0) / wc, 3), 'issue_stall': round(m.get('SQ_WAIT_INST_ANY', 0.0) / wc, 3),
                                          'issuing': round(m.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 3)}
            if k in fe and fe[k].get('FETCH_SIZE'):
                e['hbm_read_MB_per_launch'] = round(2.0 * sum(fe[k]['FETCH_SIZE']) / len(fe[k]['FETCH_SIZE']) / 1024.0, 2)
            if k in wr and wr[k].get('WRITE_SIZE'):
                e['hbm_write_MB_per_launch'] = round(sum(wr[k]['WRITE_SIZE']) / len(wr[k]['WRITE_SIZE']) / 1024.0, 2)
            if len(e) > 2:
                out['kernels'][k] = e
        json.dump(out, open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_summary.json'), 'w'), indent=1)
        if 'no-latest' not in opts and '--no-latest' not in sys.argv:
            json.dump(out, open(os.path.join(ROOT, 'profiles', 'latest_pmc_summary.json'), 'w'), indent=1)      # what bench.py reads
    tot = sum(v[1] for v in dur.values())
    for k in sorted(dur, key=lambda k: -dur[k][1])[:14]:
        e = out['kernels'].get(k, {})
        print(f'{k:34s} calls {dur[k][0]:6d}  avg {dur[k][1] / dur[k][0] / 1e3:8.1f} us  {100 * dur[k][1] / tot:5.1f} %  mfma {e.get("mfma_busy_frac", "")}'
              f'  rd {e.get("hbm_read_MB_per_launch", "")} MB  wr {e.get("hbm_write_MB_per_launch", "")} MB')


if __name__ == '__main__':
    main()
