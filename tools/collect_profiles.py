#!/usr/bin/env python3
"""Turn the files tools/measure.sh left in gpurun_out/ into the tracked summaries under profiles/:
    python tools/collect_profiles.py TAG
-> profiles/TAG_bench.json, TAG_bench_driver_cmd.json, TAG_bench_2rank_gloo_1gpu.json, TAG_bench_configs.md,
   TAG_kernel_stats_and_pmc.md, and the workloads' entries of profiles/traffic.json (HBM bytes per launch, issue fraction)."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else os.popen(f"git -C {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))} rev-parse --short HEAD").read().strip()
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
from bench import csrc_sha16   # noqa: E402  (the hash of the sources the measured library was built from: bench.py prints the same next to roofline.traffic)
CSRC_SHA = csrc_sha16()
for name in ("bench.json", "bench_driver_cmd.json", "bench_2rank_gloo_1gpu.json", "bench_2rank_north_star_gloo_1gpu.json", "bench_walk8192.json"):
    src = os.path.join(G, f"{tag}_{name}")
    if os.path.exists(src):
        lines = [l for l in open(src).read().splitlines() if l.strip().startswith("{")]
        with open(os.path.join(P, f"{tag}_{name}"), "w") as f:
            f.write(lines[-1] + "\n")
rows = []
src = os.path.join(G, f"{tag}_bench_configs.jsonl")
if os.path.exists(src):
    with open(os.path.join(P, f"{tag}_bench_configs.md"), "w") as f:
        f.write(f"# {tag} -- bench.py lines of the other workloads (tools/bench_configs.sh: --steps 1000 --warmup 100 after the 1 500-step pre-roll), MI355X, 1 GPU\n\n")
        f.write("| workload | env-steps/s (one launch per step) | ms / step | kernel ms (device timestamps over the timed launches) | one launch per 25-step segment | per 100-step segment | "
                "closed loop: PyTorch policy + rex_step | fused actor, per step | fused, 25-step segments | fused, 100-step segments |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for l in open(src):
            try:
                d = json.loads(l)
            except ValueError:
                continue
            sg = d.get("segment_launch") or {}
            sg4 = sg.get("longer_segments") or {}
            cl = d.get("closed_loop") or {}
            f.write(f"| {d['config']['workload'].split(', dt 1 ms')[0]} | {d['value'] / 1e6:.2f} M | {d['ms_per_step']:.4f} | {d['roofline']['kernel_ms']:.4f} | "
                    f"{sg.get('value', float('nan')) / 1e6:.2f} M (x{sg.get('vs_one_launch_per_step', float('nan')):.2f}) | "
                    f"{sg4.get('value', float('nan')) / 1e6:.2f} M (x{sg4.get('vs_one_launch_per_step', float('nan')):.2f}) | " +
                    " | ".join((f"{cl[k]['value'] / 1e6:.2f} M" if isinstance(cl.get(k), dict) else "--") for k in
                               ("torch_policy_per_step", "fused_per_step", "fused_segment_25", "fused_segment_100")) + " |\n")
KEYS = {"walk4096": "walk-ik/plane/base/4096", "walk262144": "walk-ik/plane/base/262144", "arm4096": "walk-ik/plane/arm/4096",
        "mixedarm2048": "mixed-ik/plane/arm/2048", "gallop8192": "gallop-ol/plane/base/8192", "turnhf4096": "turn-ik/random/base/4096",
        "poses4096": "poses-ik/plane/base/4096", "walk8192": "walk-ik/plane/base/8192", "gallop65536": "gallop-ol/plane/base/65536",
        "turnhf32768": "turn-ik/random/base/32768", "mixedarm16384": "mixed-ik/plane/arm/16384"}
ALGO = {"walk4096": 541 * 4096, "walk262144": 541 * 262144, "arm4096": 661 * 4096, "mixedarm2048": 749 * 2048, "gallop8192": 581 * 8192,
        "turnhf4096": 621 * 4096, "poses4096": 537 * 4096, "walk8192": 541 * 8192, "gallop65536": 581 * 65536, "turnhf32768": 621 * 32768,
        "mixedarm16384": 749 * 16384}
traffic_path = os.path.join(P, "traffic.json")
traffic = json.load(open(traffic_path))
traffic["_comment"] = ("HBM bytes per launch of the dominant kernel (one rex_step launch of the workload named by the key) from rocprofv3 PMC passes: FETCH_SIZE + "
                       "WRITE_SIZE, both in KiB and exact for this kernel's access pattern (profiles/r04_hbm_counter_calibration.md); issue_frac = VALU busy "
                       "fraction from the SQ pass.  Every entry names the commit whose library was measured (library_commit) and the table it comes from; "
                       "bench.py copies the entry of the workload it runs into roofline.traffic / issue_frac.")
out = [f"# {tag} -- rocprofv3 kernel trace + PMC passes of bench.py per workload (tools/profile_round.sh), MI355X\n",
       "Each workload: `rocprofv3 --kernel-trace --stats` (launches 1 550 .. 1 949 of the step kernel: the 400 timed launches of that bench.py run), then three separate `--pmc` passes "
       "(FETCH_SIZE; WRITE_SIZE; SQ_* with GRBM_GUI_ACTIVE).  FETCH_SIZE / WRITE_SIZE in KiB as gfx950 reports them (calibrated on this kernel's 4-byte-per-lane word "
       "loads: true bytes, profiles/r04_hbm_counter_calibration.md); issue fraction = 4 x SQ_ACTIVE_INST_VALU / (32 SIMDs x GRBM_GUI_ACTIVE) per shader "
       "engine: the share of SIMD cycles in which the VALU is busy (a wave64 fma keeps it busy ~2 cycles, a DPP add ~9: profiles/r04_microbench.md).\n",
       "| workload | kernel | avg us (rocprofv3, the 400 timed launches) | bench.py kernel_ms x 1000 (device timestamps, same launches) | min | max | scratch B | FETCH KiB | WRITE KiB | HBM MB / launch | algorithmic MB | ratio | VALU busy fraction | WAIT_ANY / WAVE_CYCLES |",
       "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for f in sorted(glob.glob(os.path.join(G, f"{tag}_profile_*.txt"))):
    name = os.path.basename(f)[len(tag) + 9:-4]
    txt = open(f).read()
    stats = [l for l in txt.splitlines() if l.startswith("| `_ZN3rex15rex_step")]
    js = [l for l in txt.splitlines() if l.startswith('{"') and "bench_kernel_ms_same_launches" not in l]
    same = [json.loads(l) for l in txt.splitlines() if l.startswith('{"bench_kernel_ms_same_launches"')]
    if not stats or not js:
        continue
    if json.loads(js[-1]).get("FETCH_SIZE") is None or json.loads(js[-1]).get("WRITE_SIZE") is None or json.loads(js[-1]).get("SQ_ACTIVE_INST_VALU") is None:
        print(f"{name}: a counter pass is missing (timed out?) -- entry left as it was", file=sys.stderr)
        continue
    d = json.loads(js[-1]); g = lambda k: d.get(k, {}).get("avg_steady")
    st = [c.strip() for c in stats[0].split("|")]
    mb = (g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e6   # both counters are in KiB and exact for this access pattern (profiles/r04_hbm_counter_calibration.md)
    frac = 4 * g("SQ_ACTIVE_INST_VALU") / (32 * g("GRBM_GUI_ACTIVE"))
    kern = st[1].split("rex_step_kernelI")[1].split("EEv")[0].replace("Lb0", "0").replace("Lb1", "1").replace("Li", "").replace("E", ",")
    out.append(f"| {KEYS.get(name, name)} | `<{kern}>` | {st[5]} | {same[-1]['bench_kernel_ms_same_launches'] * 1e3 if same else float('nan'):.1f} | {st[6]} | {st[7]} | {st[-2]} | {g('FETCH_SIZE'):.1f} | {g('WRITE_SIZE'):.1f} | {mb:.2f} | "
               f"{ALGO[name] / 1e6:.2f} | {mb / (ALGO[name] / 1e6):.2f} | {frac:.3f} | {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f} |")
    traffic[KEYS.get(name, name)] = {"bytes_per_launch": int(mb * 1e6), "fetch_kib": round(g("FETCH_SIZE"), 1), "write_kib": round(g("WRITE_SIZE"), 1),
                                     "issue_frac": round(frac, 4), "kernel_us_steady": float(st[5]), "source": f"profiles/{tag}_kernel_stats_and_pmc.md",
                                     "library_commit": commit, "csrc_sha16": CSRC_SHA}
with open(os.path.join(P, f"{tag}_kernel_stats_and_pmc.md"), "w") as f:
    f.write("\n".join(out) + "\n")
with open(traffic_path, "w") as f:
    json.dump(traffic, f, indent=2)
print("\n".join(out[3:]))
