#!/usr/bin/env python
"""Per-kernel summary (and optionally the timeline of the last solve) from a rocprofv3 ``--kernel-trace`` result
database (rocpd SQLite, the default output format of rocprofv3 in ROCm 7.x).

    python profiles/kernel_stats.py gpurun_out/<dir>/<name>_results.db [--timeline N] [--csv out.csv]
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
    stats = {}
    for name, s, e, *_ in rows:
        d = stats.setdefault(name, [0, 0, 1 << 62, 0])
        d[0] += 1
        d[1] += e - s
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
    total = sum(v[1] for v in stats.values())
    lines = ["name,calls,total_us,avg_us,min_us,max_us,pct"]
    for name, (n, t, lo, hi) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        short = name.replace("(anonymous namespace)::", "").split("(")[0][:70]
        lines.append(f"\"{short}\",{n},{t / 1e3:.1f},{t / n / 1e3:.2f},{lo / 1e3:.2f},{hi / 1e3:.2f},{100.0 * t / total:.1f}")
    out = "\n".join(lines)
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out + "\n")
    print(out)
    if "--timeline" in sys.argv:
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        print("\n# timeline of the last", n, "dispatches: start_us(rel) dur_us gap_us grid wg name")
        tail = rows[-n:]
        t0 = tail[0][1]
        prev = None
        for name, s, e, gx, wx, lds, vg in tail:
            gap = (s - prev) / 1e3 if prev else 0.0
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.2f} {gap:6.2f} {gx // max(wx, 1):6d} {wx:5d} {name.split('(')[0][:60]}")
            prev = e


if __name__ == "__main__":
    main()
