#!/usr/bin/env python3
"""Counter-per-byte factors of rocprofv3's FETCH_SIZE / WRITE_SIZE on this library's access pattern:
    python tools/hbm_calib.py <FETCH_SIZE results.db> <WRITE_SIZE results.db>
from a run of tools/microbench/hbm_counter_calib (known bytes per launch: 54 words x N envs x 4 B read and written).  Prints a
markdown table: per kernel and batch size the counter (KB, as reported), the known bytes, and bytes / (counter x 1000)."""
import re
import sqlite3
import sys

W = 54


def dispatches(db):
    cur = sqlite3.connect(db).cursor()
    names = dict(cur.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    pmc = dict(cur.execute("select id, name from rocpd_info_pmc"))
    rows = cur.execute("""select d.kernel_id, d.grid_size_x, d.workgroup_size_x, p.pmc_id, p.value from rocpd_pmc_event p
                          join rocpd_kernel_dispatch d on p.event_id = d.event_id order by d.event_id""").fetchall()
    out = {}
    for kid, grid, wg, pid, val in rows:
        out.setdefault((names[kid].split("(")[0], grid, pmc[pid]), []).append(val)
    return out


def main():
    print("| kernel | grid (work-items) | counter | avg per launch (KB as reported) | known bytes per launch | true bytes per reported KB |")
    print("|---|---|---|---|---|---|")
    for db in sys.argv[1:]:
        for (k, grid, c), vals in sorted(dispatches(db).items()):
            if "calib" not in k:
                continue
            steady = vals[2:] or vals
            avg = sum(steady) / len(steady)
            if "words" in k:
                m = re.search(r"ILi(\d+)E|<(\d+)>", k)
                epw = int(m.group(1) or m.group(2))
                n = grid // 64 * epw
            else:
                n = grid * 4 // W
            known = W * n * 4
            print(f"| `{k}` | {grid} | {c} | {avg:.1f} | {known} (N ~ {n}) | {known / avg:.0f} |")


if __name__ == "__main__":
    main()
