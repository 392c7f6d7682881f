#!/usr/bin/env python3
"""Copies what tools/profile_round.sh, profile_c3.sh, run_configs.py and pmc_passes.sh left under gpurun_out/ into
profiles/ (the tracked summaries) and prints the numbers the documents quote.

    python tools/collect_profiles.py [--tag r01]
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def detail(name, timed_only):
    d = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(G, name, "p*", "p_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "integrate" in k and (not timed_only or k.replace(" ", "").endswith("false>(pt::RenderArgs)")):
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not d:
        return None
    m = {k: sum(v) / len(v) for k, v in d.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    m["derived"] = {"shader_cycles": cyc, "valu_issue_utilisation": m["SQ_INSTS_VALU"] * 2 / (cyc * 1024),
                    "tcp_accesses_per_cu_over_cycles": m["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / cyc,
                    "l1_hit_rate": 1 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"],
                    "wave_cycles_waiting_on_waitcnt": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                    "wave_cycles_waiting_for_issue": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]}
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    a = ap.parse_args()
    t = a.tag
    chunks = json.load(open(os.path.join(G, "bench_1gpu.json")))["roofline"]["hbm"]["chunks_per_tile"]
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), "--tag", t, "--kernel-trace",
                           os.path.join(G, "prof_kt"), "--pmc"] + [os.path.join(G, f"prof_{k}") for k in ("fetch", "write", "sq1", "sq2", "sq3")] +
                          ["--chunks", str(chunks)], stdout=subprocess.DEVNULL)
    shutil.copy(os.path.join(G, "bench_1gpu.json"), os.path.join(P, f"{t}_bench_1gpu.json"))
    if os.path.exists(os.path.join(G, "bench_c3.json")):
        c3 = json.load(open(os.path.join(G, "bench_c3.json")))
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), "--tag", f"{t}_c3", "--spp", "1024",
                               "--kernel-trace", os.path.join(G, "c3_kt"), "--pmc"] + [os.path.join(G, f"c3_{k}") for k in ("fetch", "write", "sq")] +
                              ["--chunks", str(c3["roofline"]["hbm"]["chunks_per_tile"])], stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(G, "bench_c3.json"), os.path.join(P, f"{t}_bench_c3_1024spp.json"))
    if os.path.exists(os.path.join(G, "configs.jsonl")):
        shutil.copy(os.path.join(G, "configs.jsonl"), os.path.join(P, f"{t}_configs_one_gpu.jsonl"))
    out = {}
    for name, label, timed in (("torpmc", "Tor.obj 1920x1080x64spp (bench.py, timed kernel integrate_kernel<false,false,false>)", True),
                               ("c5pmc", "replicated scene x64, 16398 triangles, 1920x1080x8spp (tools/c5_probe.py, integrate_kernel<false,true,true>)", False)):
        m = detail(name, timed)
        if m:
            out[label] = m
    json.dump(out, open(os.path.join(P, f"{t}_pmc_detail.json"), "w"), indent=1)
    b = json.load(open(os.path.join(P, f"{t}_bench_1gpu.json")))
    h = json.load(open(os.path.join(P, f"{t}_pmc_hbm.json")))
    c = h["counters_per_launch"]
    print(f"bench: {b['value']:.0f} Msamples/s, kernel {b['roofline']['kernel_ms']:.2f} ms, frac {b['roofline']['frac']:.2f} "
          f"({b['roofline']['achieved']:.0f} TFLOP/s-eq), pcie {b['pcie_inclusive']['value']:.0f} ({b['pcie_inclusive']['ms_per_step']:.1f} ms), "
          f"e2e {b['end_to_end']['seconds']:.2f} s, cpu {b.get('cpu_baseline', {}).get('value')}")
    print(f"pmc: VALU util {h['valu_issue_utilisation']:.3f}, HBM {h['hbm_bytes_per_launch'] / 1e6:.0f} MB "
          f"(fetch {c['FETCH_SIZE'] * 1024 / 1e6:.0f}, write {c['WRITE_SIZE'] * 1024 / 1e6:.0f}), VALU instr {c['SQ_INSTS_VALU']:.3g}")
    for k, v in out.items():
        print(k[:48], {kk: round(vv, 3) for kk, vv in v["derived"].items()})
    for f in (f"{t}_kernel_stats.csv", f"{t}_c3_kernel_stats.csv"):
        if os.path.exists(os.path.join(P, f)):
            print(open(os.path.join(P, f)).read().splitlines()[1][:140])
    if os.path.exists(os.path.join(P, f"{t}_bench_c3_1024spp.json")):
        c3 = json.load(open(os.path.join(P, f"{t}_bench_c3_1024spp.json")))
        h3 = json.load(open(os.path.join(P, f"{t}_c3_pmc_hbm.json")))
        print(f"c3: {c3['value']:.0f} Msamples/s, kernel {c3['roofline']['kernel_ms']:.1f} ms, e2e {c3['end_to_end']['seconds']:.2f} s, "
              f"VALU util {h3['valu_issue_utilisation']:.3f}, HBM {h3['hbm_bytes_per_launch'] / 1e6:.0f} MB")
    if os.path.exists(os.path.join(P, f"{t}_configs_one_gpu.jsonl")):
        for line in open(os.path.join(P, f"{t}_configs_one_gpu.jsonl")):
            j = json.loads(line)
            print(j["config"], j["kernel_ms"], j.get("kernel_ms_with_statistics"), j["nominal_Msamples_per_s"], j["traced_Msamples_per_s"])


if __name__ == "__main__":
    main()
