"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db, or the *_kernel_trace.csv of --output-format csv) into a
per-kernel table.
   python tools/rocprof_summary.py gpurun_out/prof_bench/bench_results.db|..._kernel_trace.csv [out.md]"""
import sqlite3
import sys


def rows_from_csv(path):
    """rocprofv3 --output-format csv: *_kernel_trace.csv (one line per dispatch)."""
    import collections
    import csv
    g = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
        key = (r['Kernel_Name'], grid)
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        e = g.setdefault(key, {'n': 0, 'sum': 0, 'min': 1 << 62, 'max': 0, 'vgpr': 0, 'agpr': 0, 'sgpr': 0, 'lds': 0})
        e['n'] += 1
        e['sum'] += d
        e['min'] = min(e['min'], d)
        e['max'] = max(e['max'], d)
        e['vgpr'] = max(e['vgpr'], int(r.get('VGPR_Count', 0) or 0))
        e['agpr'] = max(e['agpr'], int(r.get('Accum_VGPR_Count', 0) or 0))
        e['sgpr'] = max(e['sgpr'], int(r.get('SGPR_Count', 0) or 0))
        e['lds'] = max(e['lds'], int(r.get('LDS_Block_Size', 0) or 0))
    rows = [(k[0], e['n'], e['sum'] / 1e6, e['sum'] / e['n'] / 1e3, e['min'] / 1e3, e['max'] / 1e3, e['vgpr'], e['agpr'], e['sgpr'],
             e['lds'], k[1]) for k, e in g.items()]
    rows.sort(key=lambda r: -r[2])
    return rows


def main():
    if sys.argv[1].endswith('.csv'):
        rows = rows_from_csv(sys.argv[1])
        return report(rows)
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), grid_x from kernels "
        "group by name, grid_x order by 3 desc"))
    report(rows)


def report(rows):
    tot = sum(r[2] for r in rows)
    lines = ["| kernel | grid (threads) | calls | total ms | % | avg us | min us | max us | vgpr | agpr | sgpr | lds B |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0].replace('geogcn::(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0]
        lines.append("| `%s` | %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s |" % (
            name, r[10], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
    text = "total kernel time %.2f ms over %d kernels\n\n" % (tot, len(rows)) + "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text)
    print(text)


if __name__ == '__main__':
    main()
