#!/usr/bin/env python3
"""Average PMC counter value per dispatch of the step kernel from rocprofv3 --pmc runs (rocpd sqlite databases).

    python tools/rocprof_pmc.py <FETCH_SIZE results.db> <WRITE_SIZE results.db>
gfx950: FETCH_SIZE / WRITE_SIZE are reported in kilobytes (MI355X_MICROARCH.md, HBM section: derived from the L2's
memory-side request counters; FETCH_SIZE counts a wide coalesced streaming read at half its bytes -- this kernel's state
loads are 4-byte-per-lane word loads, for which the guide gives no correction; the ratio between variants is unaffected)."""
import json
import sqlite3
import sys


def per_kernel(db):
    cur = sqlite3.connect(db).cursor()
    names = dict(cur.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
    pmc = dict(cur.execute("select id, name from rocpd_info_pmc"))
    rows = cur.execute("""select d.kernel_id, p.pmc_id, p.value from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id""").fetchall()
    out = {}
    for kid, pid, val in rows:
        out.setdefault((names[kid], pmc[pid]), []).append(val)
    return out


def main():
    res = {}
    for db in sys.argv[1:]:
        for (k, c), vals in per_kernel(db).items():
            if "rex_step_kernel" not in k:
                continue
            steady = vals[len(vals) // 2:]          # second half: past bench.py's pre-roll
            res[c] = {"kernel": k[:80], "dispatches": len(vals), "avg_all": sum(vals) / len(vals), "avg_steady": sum(steady) / len(steady)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
