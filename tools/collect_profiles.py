#!/usr/bin/env python3
"""Copies what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ (the tracked summaries) and prints the
numbers the documents quote.

    python tools/collect_profiles.py [--tag r03]

    profiles/<tag>_bench_1gpu.json        the default bench.py line
    profiles/<tag>_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the same command
    profiles/<tag>_pmc_hbm.json           the live counters bench.py collected (kernel_source_sha, kernel_ms): --pmc file reads it
    profiles/<tag>_pmc_detail.json        SQ / TCP counter passes of the timed Tor.obj kernel and of the x64 replica, with derived ratios
    profiles/<tag>_configs_one_gpu.jsonl  every BASELINE configuration on one GPU
    profiles/<tag>_mutation_sweep.jsonl   tools/mutation_sweep.py
    profiles/<tag>_phase_shares.txt       per-phase shader-clock shares (libpt_phase.so)
    profiles/<tag>_blockprof_{tor,x64}.txt  executed instructions per source region / line (tools/asm_profile.py)
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def timed_instantiation(kernel_name):
    """integrate_kernel<SKY, BIG, STATS, ENV>: the timed launches are the statistics-free ones (third argument false)."""
    m = re.search(r"integrate_kernel<(\w+),(\w+),(\w+),(\w+)(?:,(\w+))?(?:,(\w+))?>", kernel_name.replace(" ", ""))
    return bool(m) and m.group(3) == "false"


def detail(d, timed_only):
    m = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "p*", "p_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "integrate" in k and (not timed_only or timed_instantiation(k)):
                m[r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
    if not m:
        return None
    out = {}
    for k, v in m.items():
        per = collections.defaultdict(float)
        for disp, val in v:
            per[disp] += val
        out[k] = sum(per.values()) / len(per)
    cyc = out["GRBM_GUI_ACTIVE"] / 8
    out["derived"] = {"shader_cycles": cyc, "valu_issue_utilisation": out["SQ_INSTS_VALU"] * 2 / (cyc * 1024),
                      "valu_active_lane_fraction": out["SQ_THREAD_CYCLES_VALU"] / (out["SQ_INSTS_VALU"] * 64),
                      "tcp_accesses_per_cu_over_cycles": out["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256 / cyc,
                      "l1_hit_rate": 1 - out["TCP_TCC_READ_REQ_sum"] / out["TCP_TOTAL_CACHE_ACCESSES_sum"],
                      "wave_cycles_waiting_on_waitcnt": out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"],
                      "wave_cycles_waiting_for_issue": out["SQ_WAIT_INST_ANY"] / out["SQ_WAVE_CYCLES"],
                      "valu_instructions_per_wave": out["SQ_INSTS_VALU"] / out["SQ_WAVES"]}
    return out


def phase_shares(path, names):
    lines = []
    for l in open(path):
        if l.startswith("PT_PHASE_TIMERS cycles:"):
            c = [int(x) for x in l.split(":")[1].split()]
            tot = sum(c)
            lines.append("  " + ", ".join(f"{n} {100 * v / tot:.1f} %" for n, v in zip(names, c) if v))
        elif "amdgpu.ids" not in l and l.strip():
            lines.append(l.rstrip())
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r04")
    a = ap.parse_args()
    t = a.tag
    G = os.path.join(ROOT, "gpurun_out", t)
    os.makedirs(P, exist_ok=True)
    shutil.copy(os.path.join(G, "bench_1gpu.json"), os.path.join(P, f"{t}_bench_1gpu.json"))
    shutil.copy(os.path.join(G, "pmc_hbm.json"), os.path.join(P, f"{t}_pmc_hbm.json"))
    if os.path.exists(os.path.join(G, "pmc_hbm_c2_pmc_hbm.json")):      # BASELINE configs[2]: the counters of the 1024-spp launch
        shutil.copy(os.path.join(G, "pmc_hbm_c2_pmc_hbm.json"), os.path.join(P, f"{t}_c2_pmc_hbm.json"))
    for name in ("open_scene_probe.jsonl", "t_sweep.jsonl", "band_balance.jsonl"):
        if os.path.exists(os.path.join(G, name)) and os.path.getsize(os.path.join(G, name)) > 0:
            shutil.copy(os.path.join(G, name), os.path.join(P, f"{t}_{name}"))
    sky = glob.glob(os.path.join(G, "kt_sky", "**", "*kernel_stats.csv"), recursive=True)
    if sky:
        shutil.copy(max(sky, key=os.path.getmtime), os.path.join(P, f"{t}_sky_kernel_stats.csv"))
    shutil.copy(os.path.join(G, "configs.jsonl"), os.path.join(P, f"{t}_configs_one_gpu.jsonl"))
    shutil.copy(os.path.join(G, "mutation_sweep.jsonl"), os.path.join(P, f"{t}_mutation_sweep.jsonl"))
    ks = max(glob.glob(os.path.join(G, "kt", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    shutil.copy(ks, os.path.join(P, f"{t}_kernel_stats.csv"))
    out = {}
    for name, label, timed in (("pmc_tor", "Tor.obj 1920x1080x256spp (pt_render -BENCH_STEPS 1, timed kernel integrate_kernel<false,false,false,false,false,0>)", True),
                               ("pmc_x64", "replicated scene x64, 16398 triangles, 1920x1080x256spp (pt_render -BENCH_STEPS 1, timed kernel integrate_kernel<false,true,false,false,false,0>)", True),
                               ("pmc_sky", "Tor.obj without its back wall under a sky bitmap, 1920x1080x256spp (pt_render -SKYBOX -BENCH_STEPS 1, timed kernel integrate_kernel<true,false,false,false,false,0>, path regeneration)", True)):
        m = detail(os.path.join(G, name), timed)
        if m:
            out[label] = m
    json.dump(out, open(os.path.join(P, f"{t}_pmc_detail.json"), "w"), indent=1)
    if os.path.exists(os.path.join(G, "hip_startup_probe.txt")):
        head = ("# tools/hip_startup_probe.cpp, three fresh processes on the GPU box: seconds per HIP call of a program that does nothing else\n"
                "# (runtime initialisation, first stream, first kernel, pinning, teardown = real - sum), then pt_render's own phases, cold.\n")
        body = open(os.path.join(G, "hip_startup_probe.txt")).read()
        if os.path.exists(os.path.join(G, "e2e_cold_phases.txt")):
            body += "\n# pt_render --W 1920 --H 1080 -RPP 256 -TIMING 1 -FASTEXIT 1, three fresh processes (no other process on the GPU):\n" + \
                    open(os.path.join(G, "e2e_cold_phases.txt")).read()
        open(os.path.join(P, f"{t}_hip_startup_probe.txt"), "w").write(head + body)
    names = ["ray generation / loop control", "cluster + top-level tests", "tree walk (box-tree rounds)", "barycentric cull of the large class",
             "pair publication", "exact rounds (final drains)", "pre-filter + exact inside the box walk", "shading"]
    txt = ["# Per-phase shares of the waves' shader-clock cycles (diagnostic build libpt_phase.so, -DPT_PHASE_TIMERS: s_memtime stamps).",
           "# Tor.obj 1920x1080x16spp (tools/tor_probe.py):"] + phase_shares(os.path.join(G, "phase_tor.log"), names) + \
          ["# replicated scenes x64 and x195, 1920x1080x8spp (tools/c5_probe.py):"] + phase_shares(os.path.join(G, "phase_x64_x195.log"), names)
    open(os.path.join(P, f"{t}_phase_shares.txt"), "w").write("\n".join(txt) + "\n")
    # dynamic instruction profiles: counters of the instrumented code object joined with its map (tools/asm_profile.py)
    import subprocess
    import sys
    mp = os.path.join(ROOT, "path-tracing_amd", "lib", "blockprof", "map.json")
    for scene, kern in (("tor", "_ZN2pt16integrate_kernelILb0ELb0ELb0ELb0ELb0ELi0EEEvNS_10RenderArgsE"),
                        ("x64", "_ZN2pt16integrate_kernelILb0ELb1ELb0ELb0ELb0ELi0EEEvNS_10RenderArgsE")):
        cnt = os.path.join(G, f"blockprof_{scene}.{kern}.txt")
        log = os.path.join(G, f"blockprof_{scene}.log")
        if not (os.path.exists(cnt) and os.path.exists(mp) and os.path.exists(log)):
            continue
        ws = [l.split()[2] for l in open(log) if "wave_segments" in l][0]
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_profile.py"), "report", mp, cnt, "--kernel", kern,
                            "--wave-segments", ws, "--regions", os.path.join(ROOT, "tools", "asm_profile_regions.txt"),
                            "--source", os.path.join(ROOT, "path-tracing_amd", "csrc", "pt_kernels.hip"), "--top", "25"],
                           capture_output=True, text=True, check=True)
        head = (f"# Executed instructions of the timed integrator kernel, {'Tor.obj 1920x1080x8spp' if scene == 'tor' else 'x64 replica 1920x1080x4spp'}:\n"
                "# tools/asm_profile.py instruments the compiler's own assembly (one counter per straight-line run), libpt_blockprof.so\n"
                "# launches that code object; counts joined with the line tables.  The totals by kind agree with SQ_INSTS_VALU / SALU / BRANCH.\n")
        open(os.path.join(P, f"{t}_blockprof_{scene}.txt"), "w").write(head + r.stdout)
        print(r.stdout.split("\n")[0])
    b = json.load(open(os.path.join(P, f"{t}_bench_1gpu.json")))
    rf = b["roofline"]
    print(f"bench: {b['value']:.0f} Msamples/s, {b['ms_per_step']:.2f} ms/step, kernel {rf['kernel_ms']:.2f} ms, frac {rf['frac']:.3f} "
          f"({rf['achieved']:.1f} Tlane-op/s), lanes active {rf['valu_active_lane_fraction']:.3f}, traffic {rf['traffic'] / 1e6:.0f} MB "
          f"({rf['hbm']['traffic_over_algorithmic']:.2f}x), ref-eq {rf['reference_equivalent_tflops']:.0f} TFLOP/s")
    print(f"  configs3 {b['configs3_strong']['value']:.0f}, pcie {b['pcie_inclusive']['value']:.0f} ({b['pcie_inclusive']['ms_per_step']:.1f} ms), "
          f"e2e {b['end_to_end']['seconds']:.2f} s {b['end_to_end']['phases']}, cpu {b['cpu_baseline']['value']:.2f} / {b['cpu_baseline']['value_4_threads']:.2f}")
    for k, v in out.items():
        print(k[:40], {kk: round(vv, 3) for kk, vv in v["derived"].items()})
    print(open(os.path.join(P, f"{t}_kernel_stats.csv")).read()[:600])
    for line in open(os.path.join(P, f"{t}_configs_one_gpu.jsonl")):
        j = json.loads(line)
        print(j["config"], j["triangles"], j["kernel_ms"], j["nominal_Msamples_per_s"], j["traced_Msamples_per_s"], j["node_rounds_per_wave_segment"])
    print("\n".join(txt))


if __name__ == "__main__":
    main()
