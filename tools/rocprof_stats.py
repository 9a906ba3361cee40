#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --stats` run (its rocpd sqlite database) per kernel.

    python tools/rocprof_stats.py gpurun_out/<dir>/<host>/<pid>_results.db [--skip N] > profiles/<name>.md

`--skip N` leaves the first N dispatches of every kernel out of the steady-state columns (bench.py pre-rolls 1 500 steps
after a synchronous reset, during which the solver converges faster than in the stationary episode mix it reports)."""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--window", default=None, help="a:b -- steady-state columns over dispatches a..b-1 of every kernel instead of --skip")
    ap.add_argument("--csv", default=None, help="also write the table as CSV to this path (the per-kernel statistics kept under profiles/)")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    names = dict(cur.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    regs = {r[0]: r[1:] for r in cur.execute("select id, arch_vgpr_count, accum_vgpr_count, sgpr_count, group_segment_size, private_segment_size from rocpd_info_kernel_symbol")}
    per = {}
    for kid, s, e, gx, wx in cur.execute("select kernel_id, start, end, grid_size_x, workgroup_size_x from rocpd_kernel_dispatch order by start"):
        per.setdefault(kid, []).append((e - s, gx, wx))
    print("| kernel | calls | avg us (all) | steady calls | avg us | min us | max us | grid | wg | VGPR | AGPR | SGPR | LDS B | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    csv_rows = []
    for kid, rows in sorted(per.items(), key=lambda kv: -sum(r[0] for r in kv[1])):
        d = [r[0] / 1e3 for r in rows]
        if a.window:
            lo, hi = (int(v) for v in a.window.split(":"))
            st = d[lo:hi] or d
        else:
            st = d[a.skip:] if len(d) > a.skip else d
        v, ag, sg, lds, scr = regs[kid]
        print(f"| `{names[kid][:90]}` | {len(d)} | {sum(d) / len(d):.1f} | {len(st)} | {sum(st) / len(st):.1f} | {min(st):.1f} | {max(st):.1f} | "
              f"{rows[-1][1]} | {rows[-1][2]} | {v} | {ag} | {sg} | {lds} | {scr} |")
        csv_rows.append([names[kid][:90], len(d), f"{sum(d) / len(d):.1f}", len(st), f"{sum(st) / len(st):.1f}", f"{min(st):.1f}", f"{max(st):.1f}",
                         rows[-1][1], rows[-1][2], v, ag, sg, lds, scr])
    if a.csv:
        import csv
        with open(a.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow([f"# rocprofv3 --kernel-trace --stats, per kernel; 'steady' columns = dispatches {a.window or str(a.skip) + ':'} of each kernel; tools/rocprof_stats.py"])
            w.writerow(["kernel", "calls", "avg us (all)", "steady calls", "avg us", "min us", "max us", "grid", "wg", "VGPR", "AGPR", "SGPR", "LDS B", "scratch B"])
            w.writerows(csv_rows)


if __name__ == "__main__":
    main()
