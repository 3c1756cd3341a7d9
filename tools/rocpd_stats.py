"""Per-kernel duration summary from a rocprofv3 rocpd SQLite file (or several).
Usage: python tools/rocpd_stats.py <results.db> [...]   -> markdown table on stdout."""
import sqlite3
import sys


def stats(path):
    c = sqlite3.connect(path)
    suf = [r[0] for r in c.execute("select name from sqlite_master where type='table'")
           if r[0].startswith("rocpd_kernel_dispatch")][0].replace("rocpd_kernel_dispatch", "")
    q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         "sum(d.end-d.start), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), "
         "max(d.grid_size_x), max(d.workgroup_size_x) "
         "from rocpd_kernel_dispatch%s d join rocpd_info_kernel_symbol%s s on d.kernel_id = s.id "
         "group by s.kernel_name order by 6 desc" % (suf, suf))
    rows = list(c.execute(q))
    tot = sum(r[5] for r in rows) or 1
    print("| kernel | calls | avg us | min us | max us | total % | vgpr | sgpr | lds B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %d | %.2f | %.2f | %.2f | %.1f | %s | %s | %s | %s | %s |" % (
            r[0].replace(".kd", ""), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot,
            r[6], r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print("### %s" % p)
        stats(p)
