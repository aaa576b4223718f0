"""Turn rocprofv3 --pmc CSV passes (one counter set per pass) into the per-launch HBM traffic the
bench's roofline.traffic field reports.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts
TCC_EA0_RDREQ x 64 B while wide coalesced reads are 128-byte requests -> double the read side.
   python tools/pmc_summary.py gpurun_out <kernel-substring> profiles/<name>.md profiles/pmc_spmm_latest.json [state]
`state` (default: the basename of the .md without extension, e.g. r04_b_pmc_spmm) and the repository's HEAD commit are written
into the JSON, so that bench.py's `roofline.traffic_source` names the measurement it quotes."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def main():
    root, kern, out_md, out_json = sys.argv[1:5]
    state = sys.argv[5] if len(sys.argv) > 5 else os.path.splitext(os.path.basename(out_md))[0]
    try:
        commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True,
                                cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
    except OSError:
        commit = None
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, 'pmc_*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
                agg[r['Counter_Name']].append((float(r['Counter_Value']), us))
    avg = {k: sum(v[0] for v in vals) / len(vals) for k, vals in agg.items()}
    n = {k: len(vals) for k, vals in agg.items()}
    us = {k: sum(v[1] for v in vals) / len(vals) for k, vals in agg.items()}
    fetch_kb, write_kb = avg.get('FETCH_SIZE'), avg.get('WRITE_SIZE')
    res = {'kernel': kern, 'state': state, 'commit': commit, 'summary': os.path.relpath(out_md), 'counters_avg_per_launch': avg,
           'launches': n, 'avg_us_under_profiler': us}
    if fetch_kb is not None and write_kb is not None:
        res['hbm_read_bytes_per_launch'] = 2 * fetch_kb * 1024
        res['hbm_write_bytes_per_launch'] = write_kb * 1024
        res['hbm_bytes_per_launch'] = (2 * fetch_kb + write_kb) * 1024
        res['correction'] = 'read side = 2 x FETCH_SIZE (gfx950: 128-B requests tallied at 64 B)'
    if 'TCC_HIT_sum' in avg:
        res['l2_hit_rate'] = avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum'])
    json.dump(res, open(out_json, 'w'), indent=1)
    lines = ['# rocprofv3 --pmc passes, kernel `%s` (per-launch averages)\n' % kern,
             '| counter | avg per launch | launches | avg us (profiled) |', '|---|---|---|---|']
    for k in sorted(avg):
        lines.append('| %s | %.6g | %d | %.1f |' % (k, avg[k], n[k], us[k]))
    lines.append('')
    for k in ('hbm_read_bytes_per_launch', 'hbm_write_bytes_per_launch', 'hbm_bytes_per_launch', 'l2_hit_rate', 'correction'):
        if k in res:
            lines.append('* %s = %s' % (k, res[k]))
    open(out_md, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
