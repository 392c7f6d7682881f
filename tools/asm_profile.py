#!/usr/bin/env python3
"""Dynamic instruction profile of the integrator kernels, by instrumenting the COMPILER'S OWN assembly.

There is no thread-trace decoder and no PC sampling on this pool, and DESIGN.md section 7 shows the kernel's time is
proportional to the number of instructions its waves issue -- so the profile that matters is "how often does each
instruction run".  This tool produces exactly that:

    instrument  takes the device assembly of pt_kernels.hip (hipcc -S --cuda-device-only -gline-tables-only
                -DPT_BLOCK_PROFILE), and in front of every straight-line run of instructions of the chosen kernels (a run
                starts at a label or after a branch) inserts four instructions that add 1 to a per-wave LDS counter;
                the counters are flushed to RenderArgs::blockprof at s_endpgm.  It writes the instrumented assembly
                and a map  run id -> instructions of the run with their source lines.
    report      joins the counters a run of libpt_blockprof.so wrote ($PT_BLOCKPROF_OUT.<kernel>.txt) with that map:
                executed instructions per source line / per region, per wave-segment.

The inserted code uses registers the kernel does not (v110.. , s[100:101]) and saves / restores EXEC; it changes no
flag the surrounding code reads (s_mov does not write SCC; VCC is untouched after the prologue).  Waits only get longer
(an extra LDS operation in flight).  `make -C path-tracing_amd/csrc blockprof` builds the instrumented code object and
libpt_blockprof.so, tools/blockprof_run.py renders a frame through them, tools/collect_profiles.py writes the reports.
"""
import argparse
import collections
import json
import re
import sys

LDS_BASE = 8192          # counters live above everything the kernels allocate themselves (<= 7232 bytes)
MAX_RUNS = 1024
INSTR = re.compile(r"\s+((?:v|s|ds|global|buffer|flat|scratch)_\w+)(.*)")


def instrument(args):
    kernels = args.kernels.split(",")
    out, maps = [], {}
    cur_kernel, run_id, cur_loc, runs = None, 0, None, None
    lines = open(args.asm).read().split("\n")
    i = 0

    def counter(k):
        return [f"\ts_mov_b64 s[100:101], exec", f"\ts_mov_b64 exec, 1", f"\tds_add_u32 v110, v111 offset:{LDS_BASE + 4 * k}",
                f"\ts_mov_b64 exec, s[100:101]"]

    def new_run():
        nonlocal run_id
        if run_id >= MAX_RUNS:
            raise SystemExit("too many runs")
        out.extend(counter(run_id))
        runs.append([])
        run_id += 1

    while i < len(lines):
        line = lines[i]
        i += 1
        m = re.match(r"(_ZN2pt16integrate_kernel\w+):", line)
        if m and m.group(1) in kernels:
            cur_kernel, run_id, runs = m.group(1), 0, []
            maps[cur_kernel] = runs
            out.append(line)
            # prologue: v110 = LDS address 0, v111 = 1, v112 = lane * 4, v[114:115] = blockprof + lane * 4; counters zeroed
            out += ["\tv_mov_b32 v110, 0", "\tv_mov_b32 v111, 1", f"\ts_load_dwordx2 s[100:101], s[0:1], {args.kernarg_offset}",
                    "\tv_lshlrev_b32 v112, 2, v0"]
            for j in range(MAX_RUNS // 64):
                out.append(f"\tds_write_b32 v112, v110 offset:{LDS_BASE + 256 * j}")
            out += ["\ts_waitcnt lgkmcnt(0)", "\tv_mov_b32 v114, s100", "\tv_mov_b32 v115, s101",
                    "\tv_add_co_u32 v114, vcc, v114, v112", "\tv_addc_co_u32 v115, vcc, 0, v115, vcc"]
            new_run()
            continue
        if cur_kernel is None:
            out.append(line)
            continue
        if line.startswith(".Lfunc_end"):
            cur_kernel = None
            out.append(line)
            continue
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur_loc = (int(m.group(1)), int(m.group(2)))
            out.append(line)
            continue
        if re.match(r"\.LBB\d+_\d+:", line):
            out.append(line)
            new_run()
            continue
        m = INSTR.match(line)
        if not m:
            out.append(line)
            continue
        op = m.group(1)
        if op == "s_endpgm":
            out += ["\ts_mov_b64 exec, -1", "\ts_waitcnt vmcnt(0) lgkmcnt(0)"]
            for j in range(MAX_RUNS // 64):
                if j and j % 16 == 0:
                    out += ["\tv_add_co_u32 v114, vcc, 0x1000, v114", "\tv_addc_co_u32 v115, vcc, 0, v115, vcc"]
                out += [f"\tds_read_b32 v116, v112 offset:{LDS_BASE + 256 * j}", "\ts_waitcnt lgkmcnt(0)",
                        f"\tglobal_atomic_add v[114:115], v116, off offset:{256 * (j % 16)}"]
            out += ["\ts_waitcnt vmcnt(0)", line]
            continue
        runs[-1].append([op, cur_loc[0] if cur_loc else -1, cur_loc[1] if cur_loc else 0, line.split(";")[0].strip()[len(op):].strip()])
        out.append(line)
        if op.startswith("s_cbranch") or op == "s_branch":
            new_run()
    text = "\n".join(out)
    # kernel descriptors: room for the extra registers and the counters' LDS
    def patch(block):
        name = re.search(r"\.amdhsa_kernel (\S+)", block.group(0)).group(1)
        if name not in kernels:
            return block.group(0)
        b = block.group(0)
        b = re.sub(r"\.amdhsa_group_segment_fixed_size \d+", f".amdhsa_group_segment_fixed_size {LDS_BASE + 4 * MAX_RUNS}", b)
        b = re.sub(r"\.amdhsa_next_free_vgpr \d+", ".amdhsa_next_free_vgpr 120", b)
        b = re.sub(r"\.amdhsa_accum_offset \d+", ".amdhsa_accum_offset 120", b)
        b = re.sub(r"\.amdhsa_next_free_sgpr \d+", ".amdhsa_next_free_sgpr 102", b)
        return b
    text = re.sub(r"\.amdhsa_kernel .*?\.end_amdhsa_kernel", patch, text, flags=re.S)
    # the metadata (YAML note) carries the same numbers; the loader trusts the descriptor, but keep them consistent
    for k in kernels:
        def meta(mm):
            b = mm.group(0)
            b = re.sub(r"\.group_segment_fixed_size: \d+", f".group_segment_fixed_size: {LDS_BASE + 4 * MAX_RUNS}", b)
            b = re.sub(r"\.vgpr_count:\s+\d+", ".vgpr_count:     120", b)
            b = re.sub(r"\.sgpr_count:\s+\d+", ".sgpr_count:     108", b)
            return b
        text = re.sub(r"- \.agpr_count:.*?\.symbol:\s+" + re.escape(k) + r"\.kd", meta, text, flags=re.S)
    open(args.out, "w").write(text)
    files = {}
    for mm in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', open(args.asm).read()):
        files[int(mm.group(1))] = mm.group(3) or mm.group(2)
    json.dump({"runs": maps, "files": files}, open(args.map, "w"))
    for k, r in maps.items():
        print(f"{k}: {len(r)} runs, {sum(len(x) for x in r)} instructions", file=sys.stderr)


def kind(op):
    if op in ("s_nop",):
        return "nop"
    if op == "s_waitcnt":
        return "wait"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


# SIMD cycles one wave-instruction keeps its pipe busy, from tools/issue_cost.hip on an MI355X (profiles/r02_issue_cost.txt):
# the plain fp32 / integer-add / logic operations run a wave in 2 cycles when every operand is a VGPR, a literal or an
# inline constant; with a scalar-register operand, as DPP / SDWA, and for everything else (min / max, compares, selects,
# conversions, shifts left, three-operand integer forms, 64-bit and fp64, integer multiplies, lane access) it is 4;
# reciprocal / square root 8.  Scalar instructions take 4 cycles on the scalar pipe, which runs beside the vector pipe.
FAST_VALU = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32",
             "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32",
             "v_mul_lo_u16", "v_mul_legacy_f32"}
SLOW8_VALU = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}
SLOW16_VALU = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}


def pipe_cycles(op, operands):
    """(pipe, cycles) of one wave-instruction."""
    k = kind(op)
    if k == "valu":
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        if base in SLOW8_VALU:
            return "valu", 8.0
        if base in SLOW16_VALU:
            return "valu", 16.0
        scalar_operand = re.search(r"(?<![\w.])(s\d+\b|s\[\d+:\d+\]|(?:vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|ttmp\d+)\b)", operands) is not None
        if base in FAST_VALU and not scalar_operand and not op.endswith(("_dpp", "_sdwa")) and "row_" not in operands and "sel:" not in operands:
            return "valu", 2.0
        return "valu", 4.0
    if k in ("salu", "branch"):
        return "salu", 4.0
    if k in ("nop", "wait"):
        return "salu", 0.7
    if k == "lds":
        return "lds", 17.0 if "write" in op or "add" in op or "min" in op or "max" in op else 9.0
    return k, 4.0


def report(args):
    mp = json.load(open(args.map))
    runs = mp["runs"][args.kernel]
    files = {int(k): v for k, v in mp["files"].items()}
    cnt = {}
    for line in open(args.counters):
        a, b = line.split()
        cnt[int(a)] = int(b)
    src = {}
    per_line = collections.Counter()
    per_kind = collections.Counter()
    per_op = collections.Counter()
    total = 0
    line_cycles = collections.Counter()      # vector-pipe cycles per source line
    pipe_total = collections.Counter()
    scalar_operand_valu = 0
    for k, ins in enumerate(runs):
        c = cnt.get(k, 0)
        for rec in ins:
            op, f, l = rec[0], rec[1], rec[2]
            per_line[(f, l)] += c
            per_kind[kind(op)] += c
            per_op[op] += c
            total += c
            if len(rec) > 3:
                pipe, cyc = pipe_cycles(op, rec[3])
                pipe_total[pipe] += c * cyc
                if pipe == "valu":
                    line_cycles[(f, l)] += c * cyc
                    base = re.sub(r"_(e32|e64)$", "", op)
                    if cyc == 4.0 and base in FAST_VALU:
                        scalar_operand_valu += c
    ws = args.wave_segments or 1
    print(f"kernel {args.kernel}: {total} instructions executed, {total / ws:.1f} per wave-segment ({ws} wave-segments)")
    print("by kind (per wave-segment): " + "  ".join(f"{k} {v / ws:.1f}" for k, v in per_kind.most_common()))
    print("top opcodes: " + "  ".join(f"{k} {v / ws:.1f}" for k, v in per_op.most_common(25)))
    if pipe_total:
        print("pipe cycles per wave-segment (cost table of tools/issue_cost.hip): " + "  ".join(f"{k} {v / ws:.0f}" for k, v in pipe_total.most_common())
              + f"   [fp32 / add / logic instructions slowed to 4 cycles by a scalar operand, DPP or SDWA: {scalar_operand_valu / ws:.1f} per wave-segment]")
    regions = []
    if args.regions:
        src_lines = open(args.source).read().split("\n")
        at = 0
        for r in open(args.regions):
            r = r.strip()
            if not r or r.startswith("#"):
                continue
            name, rx = [x.strip() for x in r.split("|", 1)]
            for i in range(at, len(src_lines)):
                if re.search(rx, src_lines[i]):
                    regions.append([i + 1, len(src_lines) + 1, name])
                    if len(regions) > 1:
                        regions[-2][1] = i
                    at = i + 1
                    break
            else:
                raise SystemExit(f"region '{name}': no line after {at} matches {rx}")
    per_region = collections.Counter()
    other_files = collections.Counter()
    main_name = args.source.split("/")[-1] if args.regions else ""
    for (f, l), c in per_line.items():
        fname = (files.get(f, "?") if f >= 0 else "?").split("/")[-1]
        if regions and fname == main_name:
            if l == 0:
                per_region["(no source line: compiler-generated, inlined helpers without locations)"] += c
                continue
            for a, b, name in regions:
                if a <= l <= b:
                    per_region[name] += c
                    break
            else:
                per_region["(before the first region)"] += c
        else:
            other_files[fname] += c
    unassigned = other_files
    region_cycles = collections.Counter()
    main_cycles_total = sum(line_cycles.values()) or 1
    for (f, l), cy in line_cycles.items():
        fname = (files.get(f, "?") if f >= 0 else "?").split("/")[-1]
        if regions and fname == main_name and l != 0:
            for a, b, name in regions:
                if a <= l <= b:
                    region_cycles[name] += cy
                    break
        elif regions and fname != main_name:
            region_cycles["[" + fname + "]"] += cy
        elif regions:
            region_cycles["(no source line: compiler-generated, inlined helpers without locations)"] += cy
    if regions:
        print("\nby region (instructions per wave-segment, share of instructions; vector-pipe cycles per wave-segment, share of those):")
        for name, c in per_region.most_common():
            cy = region_cycles.get(name, 0)
            print(f"  {name:34s} {c / ws:8.1f}  {100.0 * c / total:5.1f} %   {cy / ws:8.0f}  {100.0 * cy / main_cycles_total:5.1f} %")
        for fname, c in other_files.most_common():
            cy = region_cycles.get("[" + fname + "]", 0)
            print(f"  {'[' + fname + ']':34s} {c / ws:8.1f}  {100.0 * c / total:5.1f} %   {cy / ws:8.0f}  {100.0 * cy / main_cycles_total:5.1f} %")
    print("\nhottest source lines:")
    for (f, l), c in per_line.most_common(args.top):
        fname = files.get(f, "?") if f >= 0 else "?"
        print(f"  {fname.split('/')[-1]}:{l:<5d} {c / ws:8.1f}  {100.0 * c / total:5.1f} %   {line_cycles.get((f, l), 0) / ws:8.0f} cycles")
    if args.runs_out:
        with open(args.runs_out, "w") as fp:
            for k, ins in enumerate(runs):
                c = cnt.get(k, 0)
                lines = collections.Counter((r[1], r[2]) for r in ins)
                top = " ".join(f"{l}x{n}" for (f, l), n in lines.most_common(5) if f == 0)
                fp.write(f"{k:4d} exec/ws {c / ws:8.3f} n={len(ins):4d} instr/ws {c * len(ins) / ws:8.1f} | {top}\n")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("instrument")
    a.add_argument("asm"); a.add_argument("out"); a.add_argument("map")
    a.add_argument("--kernels", required=True)
    a.add_argument("--kernarg-offset", type=int, required=True)
    b = sub.add_parser("report")
    b.add_argument("map"); b.add_argument("counters")
    b.add_argument("--kernel", required=True)
    b.add_argument("--wave-segments", type=float, default=0)
    b.add_argument("--regions", default="")
    b.add_argument("--source", default="path-tracing_amd/csrc/pt_kernels.hip")
    b.add_argument("--top", type=int, default=40)
    b.add_argument("--runs-out", default="")
    args = ap.parse_args()
    (instrument if args.cmd == "instrument" else report)(args)


if __name__ == "__main__":
    main()
