#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the rocprofv3 PMC passes of tools/gpu_pmc.sh (FETCH_SIZE / WRITE_SIZE, separate
passes).  Units and the gfx950 correction follow MI355X_MICROARCH.md "HBM": both counters are in KiB per dispatch,
and FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so it is doubled."""
import csv, glob, hashlib, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
root = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
def maxval(sub, counter):
    vals = []
    for f in glob.glob(str(root / sub / "**" / "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            # the headline launch only: homogeneous batch, parity-check stop (bench.py's chain_roofline leg runs other kernels
            # whose argument list mentions ldpc_dec_args)
            if row["Kernel_Name"].startswith("void ldpc_dec_fast_kernel<false, false") and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return max(vals)   # the fixed-work launches (9 passes) are the largest
model = sorted(Path("profiles").glob("r0*/valu_issue_model.json"))[-1]  # the newest round's
model_avg_ns = json.loads(model.read_text())["avg_ns_per_valu_wave_inst"] if model.exists() else None
fetch_kib, write_kib = maxval("pmc_fetch", "FETCH_SIZE"), maxval("pmc_write", "WRITE_SIZE")
valu = maxval("pmc_sq1", "SQ_INSTS_VALU")
out = {"ldpc_dec_bg1_z384_r13_b1024_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024),
       "fetch_size_kib_raw": fetch_kib, "fetch_bytes_corrected_x2": int(2 * fetch_kib * 1024),
       "write_size_kib": write_kib, "write_bytes": int(write_kib * 1024),
       "compulsory_bytes_per_launch": 1024 * (68 * 384 + 22 * 384 // 8),
       "valu_wave_insts_per_launch": int(valu),
       # mean issue time per VALU wave-instruction per SIMD: opcode histogram of the kernel's bodies (disassembly) x the
       # per-opcode issue times of profiles/r01/valu_rate_ubench.txt, weighted by the code's task structure
       # (tools/valu_issue_model.py -> profiles/rNN/valu_issue_model.json); not a hand-entered figure any more
       "valu_avg_ns_per_wave_inst_per_simd": model_avg_ns,
       # what the counters were collected on (bench.py compares this with what it runs and says `stale` when they differ)
       "measured_on": __import__("bench").build_identity(__import__("openairinterface5g_amd")),
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes on bench.py (1024 x BG1 Zc=384 R13, 9 passes)"}
Path("profiles").mkdir(exist_ok=True)
Path("profiles/hbm_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
print(out)
