#!/usr/bin/env python3
"""Where the fast decoder's time goes, derived from files instead of a hand-entered constant.

  1. tools/model/cn_probe.hip is compiled to gfx950 assembly (same flags as the library): one kernel per instantiation of
     the per-item bodies -- ldpc_fast_cn<D, EXT, MODE>, ldpc_fast_bn with loop bound M -- whose opcode histograms are read
     off the disassembly (the library inlines the same code).
  2. tools/model/tasks.c prints the task structure of the code (check-node tasks per degree, bit-node tasks per bound).
  3. Every VALU opcode is priced with the issue time measured for it by tools/ubench/valu_rate.hip
     (profiles/r01/valu_rate_ubench.txt: ns per wave-instruction per SIMD), packed-16 / perm / alignbyte / 32-bit min-max
     at ~1.8 ns, plain add / sub / xor / and / shift at ~1.1 ns.
  4. Sum over the tasks of a pass, times passes and blocks, divided by the SIMDs a block's waves share = the time the
     VALU pipes need.  Cross-check: the predicted VALU instruction count against the SQ_INSTS_VALU counter of the same
     launch (profiles/hbm_traffic.json); binding fraction = VALU issue time / measured kernel time.

  python tools/valu_issue_model.py [--kernel-ms 0.508] [--out profiles/r02/valu_issue_model.json]
"""
import argparse
import collections
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "openairinterface5g_amd" / "csrc"

# opcode -> row of the micro-benchmark table (anything not listed: priced as a plain 32-bit op, v_xor_b32's row)
UBENCH_ROW = {
    "v_pk_min_u16": "pk_min_i16", "v_pk_max_u16": "pk_max_i16", "v_pk_min_i16": "pk_min_i16", "v_pk_max_i16": "pk_max_i16",
    "v_pk_add_u16": "pk_add_u16", "v_pk_add_i16": "pk_add_u16", "v_pk_sub_i16": "pk_sub_i16", "v_pk_sub_u16": "pk_sub_i16",
    "v_pk_ashrrev_i16": "pk_ashr_i16", "v_pk_lshlrev_b16": "pk_lshl_b16", "v_pk_lshrrev_b16": "pk_lshl_b16",
    "v_pk_mul_lo_u16": "pk_mul_lo", "v_pk_mad_i16": "pk_mad_i16", "v_pk_mad_u16": "pk_mad_i16",
    "v_pk_maximum3_f16": "pk_min_i16", "v_pk_max_f16": "pk_min_i16",   # measured at the packed-min rate (tools/ubench/pk_max3_f16.hip)
    "v_perm_b32": "perm_b32", "v_alignbyte_b32": "alignbyte", "v_alignbit_b32": "alignbit",
    "v_min_u32": "min_u32", "v_max_u32": "min_u32", "v_min_i32": "min_i32", "v_max_i32": "max_i32",
    "v_min3_u32": "min3_u32", "v_med3_i32": "med3_i32", "v_med3_u32": "med3_i32",
    "v_bfe_u32": "bfe_i32", "v_bfe_i32": "bfe_i32", "v_bfi_b32": "bfi_b32", "v_and_or_b32": "and_or", "v_lshl_or_b32": "lshl_or",
    "v_lshl_add_u32": "lshl_or", "v_add_lshl_u32": "lshl_or", "v_add3_u32": "and_or", "v_xad_u32": "and_or", "v_or3_b32": "and_or",
    "v_cndmask_b32": "min_u32",      # (the table's cndmask row measures a VCC dependency chain, not the issue rate)
    "v_mul_u32_u24": "mul_u24", "v_mul_lo_u32": "mul_lo_u32", "v_mul_hi_u32": "mul_hi_u32", "v_mad_u32_u24": "mul_u24",
    "v_mad_u64_u32": "mul_lo_u32", "v_mov_b32_dpp": "mov_dpp", "v_sad_u8": "sad_u8",
    "v_add_u32": "add_u32", "v_add_co_u32": "add_u32", "v_addc_co_u32": "add_u32", "v_sub_u32": "sub_u32", "v_subrev_u32": "sub_u32",
    "v_sub_co_u32": "sub_u32", "v_and_b32": "and_b32", "v_or_b32": "and_b32", "v_xor_b32": "xor_b32", "v_not_b32": "xor_b32",
    "v_lshrrev_b32": "lshrrev", "v_lshlrev_b32": "lshrrev", "v_ashrrev_i32": "ashrrev", "v_mov_b32": "xor_b32",
}


def ubench_table():
    t = {}
    for line in (ROOT / "profiles" / "r01" / "valu_rate_ubench.txt").read_text().splitlines():
        m = re.match(r"(\S+)\s+blocks/CU=\d+\s+[\d.]+ ms\s+([\d.]+) ns/wave-instr/SIMD", line)
        if m:
            t[m.group(1)] = float(m.group(2))
    return t


def build_asm(tmp):
    s = Path(tmp) / "cn_probe.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", str(CSRC), "-S", "--cuda-device-only",
                    str(ROOT / "tools" / "model" / "cn_probe.hip"), "-o", str(s)], check=True, capture_output=True)
    exe = Path(tmp) / "tasks"
    subprocess.run(["gcc", "-O1", "-I", str(CSRC), str(ROOT / "tools" / "model" / "tasks.c"), str(CSRC / "ldpc_graph.c"), "-o", str(exe)], check=True)
    return s.read_text(), exe


def kernel_histograms(asm):
    """{kernel name: Counter(opcode)} for the straight-line bodies; loops (the bit-node gather with a run-time trip count do
    not occur: every probe has compile-time bounds)."""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(probe_\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = collections.Counter()
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = line.strip()
        if not t or t[0] in ";." or ":" in t.split()[0]:
            continue
        op = t.split()[0]
        if re.fullmatch(r"[a-z][a-z0-9_]+", op):
            out[cur][op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "_dpp")] += 1
    return out


def price(hist, table):
    ns = n_valu = n_lds = n_other = 0
    by_class = collections.Counter()
    for op, c in hist.items():
        if op.startswith("v_") and not op.startswith("v_readlane") and not op.startswith("v_writelane") and not op.startswith("v_readfirstlane"):
            row = UBENCH_ROW.get(op, "xor_b32")
            ns += c * table[row]
            n_valu += c
            by_class["slow (~1.8 ns: packed-16, perm, alignbyte, 32-bit min/max, mul, 3-operand)" if table[row] > 1.5 else
                     "fast (~1.1 ns: add, sub, xor, and, or, shift, mov)"] += c
        elif op.startswith("ds_"):
            n_lds += c
        else:
            n_other += c
    return ns, n_valu, n_lds, n_other, by_class


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel-ms", type=float, default=None, help="measured duration of the 1024-block launch (default: profiles)")
    ap.add_argument("--blocks", type=int, default=1024)
    ap.add_argument("--passes", type=int, default=9)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    table = ubench_table()
    with tempfile.TemporaryDirectory() as tmp:
        asm, tasks_exe = build_asm(tmp)
        tasks = json.loads(subprocess.run([str(tasks_exe), "1", "384", "13"], check=True, capture_output=True, text=True).stdout)
    hists = kernel_histograms(asm)
    # a probe kernel = the body + ~12 instructions of argument handling; measured on the smallest probes and subtracted
    per_kind = {}
    for name, h in hists.items():
        ns, nv, nl, no, cls = price(h, table)
        per_kind[name] = dict(ns=ns, valu=nv, lds=nl, other=no, cls=cls)
    tot_ns = tot_valu = tot_lds = 0.0
    cls_tot = collections.Counter()
    detail = []
    TASK_OVERHEAD_VALU = 14          # queue draw, task record, item index, row record, syndrome mask (ldpc_dec_fast_block.h)
    for t in tasks["cn_tasks"]:
        k = per_kind["probe_cn_19pair" if t.get("pair") else (f"probe_cn2_{t['deg']}" if t.get("double") else
                                                               f"probe_cn_{t['deg']}_{t['ext']}_{2 if t['deg'] == 19 else 0}")]
        tot_ns += k["ns"] + TASK_OVERHEAD_VALU * table["xor_b32"]
        tot_valu += k["valu"] + TASK_OVERHEAD_VALU
        tot_lds += k["lds"]
        cls_tot.update(k["cls"])
    cn_ns = tot_ns
    for t in tasks["bn_tasks"]:
        k = per_kind[f"probe_bn_{t['maxdeg']}"]
        tot_ns += k["ns"] + TASK_OVERHEAD_VALU * table["xor_b32"]
        tot_valu += k["valu"] + TASK_OVERHEAD_VALU
        tot_lds += k["lds"]
        cls_tot.update(k["cls"])
    for name in sorted(per_kind):
        k = per_kind[name]
        detail.append({"body": name, "valu": k["valu"], "lds": k["lds"], "valu_issue_ns": round(k["ns"], 1)})
    traffic = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())
    measured_valu = traffic.get("valu_wave_insts_per_launch")
    kernel_ms = args.kernel_ms
    if kernel_ms is None:
        kernel_ms = 0.513          # profiles/r01/fast_kernel_stats.csv AverageNs of ldpc_dec_fast_kernel<1024>
    waves_per_simd_share = 4         # a block's 16 waves sit on the 4 SIMDs of one CU: 4 issue ports share the block's work
    per_block_pass_us = tot_ns / 1e3 / waves_per_simd_share
    launch_ms = per_block_pass_us * args.passes * 1e-3 * (args.blocks / 256.0)   # 256 CUs, one block per CU at a time
    predicted_valu = tot_valu * args.passes * args.blocks
    res = {
        "code": "BG1 Zc=384 R=1/3", "blocks": args.blocks, "passes": args.passes,
        "valu_wave_insts_per_block_pass_model": tot_valu, "lds_insts_per_block_pass_model": tot_lds,
        "valu_wave_insts_per_launch_model": predicted_valu, "valu_wave_insts_per_launch_measured_SQ_INSTS_VALU": measured_valu,
        "model_over_measured": predicted_valu / measured_valu if measured_valu else None,
        "valu_issue_ns_per_block_pass_per_simd": per_block_pass_us * 1e3,
        "avg_ns_per_valu_wave_inst": tot_ns / tot_valu,
        "check_node_share_of_valu_issue": cn_ns / tot_ns,
        "valu_issue_ms_per_launch": launch_ms, "kernel_ms_measured": kernel_ms,
        "binding_fraction_valu_issue": launch_ms / kernel_ms,
        "opcode_classes": {k: int(v) for k, v in cls_tot.items()},
        "bodies": detail,
        "inputs": ["tools/model/cn_probe.hip (disassembly)", "tools/model/tasks.c (task structure)", "profiles/r01/valu_rate_ubench.txt (issue times)",
                   "profiles/hbm_traffic.json (SQ_INSTS_VALU of the same launch)"],
    }
    txt = json.dumps(res, indent=1)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(txt + "\n")
    print(txt)


if __name__ == "__main__":
    sys.exit(main())
