"""Summarise a rocprofv3 (--kernel-trace --stats) rocpd sqlite database into a text table for profiles/.

usage: python tools/prof_summary.py gpurun_out/prof1/r1_results.db "command line that was profiled" > profiles/xyz.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary\n# command: {sys.argv[2] if len(sys.argv) > 2 else '?'}")
    print(f"# source db: {sys.argv[1]}\n")
    print(f"{'kernel':<78} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}")
    for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print(f"{name[:78]:<78} {calls:>7} {total / 1e3:>10.3f} {avg:>10.2f} {pct:>6.2f}")
    print("\n# per launch geometry (grid is in work-items; duration in us)")
    print(f"{'kernel':<60} {'grid':>10} {'lds':>7} {'vgpr':>5} {'agpr':>5} {'calls':>6} {'avg_us':>10} {'min_us':>10}")
    q = ("select name, grid_x, lds_size, vgpr_count, accum_vgpr_count, count(*), avg(duration), min(duration) from kernels "
         "group by name, grid_x, lds_size order by sum(duration) desc limit 40")
    for r in cur.execute(q):
        print(f"{r[0][:60]:<60} {r[1]:>10} {r[2]:>7} {r[3]:>5} {r[4]:>5} {r[5]:>6} {r[6] / 1e3:>10.2f} {r[7] / 1e3:>10.2f}")


if __name__ == "__main__":
    main()
