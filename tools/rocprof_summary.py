"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a per-kernel table.
   python tools/rocprof_summary.py gpurun_out/prof_bench/bench_results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), grid_x from kernels "
        "group by name, grid_x order by 3 desc"))
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
