"""Per-case L2 / fabric counters of tools/spmm_locality.py run under rocprofv3 --pmc (separate passes for TCC_HIT/MISS,
FETCH_SIZE, WRITE_SIZE; csv output).  Launches of spmm_rows_kernel are attributed to the cases in order (each case =
reps + 3 launches).  gfx950 correction (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE KB.
   python tools/pmc_locality_summary.py <dir with the pmc passes> gpurun_out/spmm_locality.json out.md"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    root, cases_json, out_md = sys.argv[1:4]
    cases = json.load(open(cases_json))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if 'spmm_rows_kernel' in r['Kernel_Name']]
        by_counter = collections.defaultdict(list)
        for r in rows:
            by_counter[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
        for cn, vals in by_counter.items():
            vals.sort()
            k = 0
            for c in cases:
                n = c['launches']
                per[c['case']][cn] += [v for _, v in vals[k:k + n]]
                k += n
    lines = ['| case | ms | alg GB/s | frac of 8 TB/s | L2 hit rate | read GB (2 x FETCH_SIZE) | write GB | traffic / algorithmic | edges inside an L2 window |',
             '|---|---|---|---|---|---|---|---|---|']
    for c in cases:
        p = per[c['case']]
        avg = {k: sum(v) / len(v) for k, v in p.items() if v}
        hit = avg.get('TCC_HIT_sum')
        miss = avg.get('TCC_MISS_sum')
        rd = 2 * avg['FETCH_SIZE'] * 1024 if 'FETCH_SIZE' in avg else None
        wr = avg['WRITE_SIZE'] * 1024 if 'WRITE_SIZE' in avg else None
        alg = 8 * c['nnz'] + 4 * 440001 + 8 * 440000 * 300
        c.update(l2_hit_rate=None if hit is None else hit / (hit + miss), read_bytes=rd, write_bytes=wr,
                 traffic_ratio=None if rd is None or wr is None else (rd + wr) / alg)
        f = lambda v, fmt: '-' if v is None else fmt % v
        lines.append('| %s | %.3f | %.0f | %.3f | %s | %s | %s | %s | %.1f %% |' % (
            c['case'], c['ms'], c['alg_GBps'], c['frac_of_8TBps'], f(c['l2_hit_rate'], '%.3f'), f(None if rd is None else rd / 1e9, '%.2f'),
            f(None if wr is None else wr / 1e9, '%.2f'), f(c['traffic_ratio'], '%.2f'), 100 * c['edges_within_L2_window']))
    text = '\n'.join(lines) + '\n'
    open(out_md, 'w').write(text)
    json.dump(cases, open(out_md.replace('.md', '.json'), 'w'), indent=1)
    print(text)


if __name__ == '__main__':
    main()
