#!/usr/bin/env python3
"""A/B timing of decoder builds / tuning knobs on one box: every variant runs in its own process (fresh library, its own
environment) on the same seeded inputs -- the headline fixed-work launch (BG1 Zc=384 R=1/3, 1024 blocks, all 9 passes),
the operating point (Es/N0 = 1 dB, early stop) and optionally other codes.

  python tools/ab_fast.py base:NRLDPC_HIP_LIB=.../libldpc_hip_base.so new: nosplit:NRLDPC_HIP_BN_SPLIT=0
  python tools/ab_fast.py --child            (internal)
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CODES = [(1, 384, 13), (1, 384, 23), (1, 384, 89), (1, 192, 13), (2, 208, 15), (2, 384, 15), (2, 384, 23)]
if os.environ.get("AB_CODES"):          # e.g. AB_CODES=1:352:13,1:320:23
    CODES = [tuple(int(x) for x in c.split(":")) for c in os.environ["AB_CODES"].split(",")]


def child():
    import numpy as np
    import torch
    import openairinterface5g_amd as pkg
    m = pkg.ldpc
    pkg.LDPCinit()
    res = {}
    n = 1024
    for BG, Z, R in (CODES[:1] if os.environ.get("AB_ONLY_R13") else CODES):
        g = torch.Generator(device="cuda").manual_seed(7)
        ncol = m.NCOLS[(BG, R)]
        K = (22 if BG == 1 else 10) * Z
        info = torch.randint(0, 256, (n, K // 8), dtype=torch.uint8, device="cuda", generator=g)
        coded = torch.empty((n, (66 if BG == 1 else 50) * Z), dtype=torch.uint8, device="cuda")
        pkg.encode_batch_device(BG, Z, info, coded)
        coded = coded[:, :(ncol - 2) * Z]
        for name, snr in (("fixed", -12.0), ("op", 1.0 if R in (13, 15) else 4.0)):
            sigma = 1.0 / np.sqrt(2.0 * 10.0 ** (snr / 10.0))
            y = 1.0 - 2.0 * coded.float() + sigma * torch.randn(coded.shape, device="cuda", generator=g)
            llr = torch.zeros((n, ncol * Z), dtype=torch.int8, device="cuda")
            llr[:, 2 * Z:] = torch.clamp(torch.floor(y / (sigma / 16.0)), -128, 127).to(torch.int8)
            out = torch.zeros((n, m.out_bytes(BG, Z, R)), dtype=torch.uint8, device="cuda")
            it = torch.zeros(n, dtype=torch.int32, device="cuda")
            for _ in range(80):    # (the GPU's clocks ramp for ~30 ms after idling: profiles/r06/clock_ramp.txt)
                pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
            torch.cuda.synchronize()
            ev[0].record()
            for i in range(40):
                pkg.decode_batch_device(BG, Z, R, llr, out, it, numMaxIter=8)
                ev[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
            res[f"BG{BG} Z{Z} R{R} {name}"] = dict(ms_median=ms[20], ms_min=ms[0], passes=float(it.float().mean()),
                                                   checksum=int(out.to(torch.int64).sum()) + int(it.sum()))
    print(json.dumps(res))


def main():
    variants = sys.argv[1:] or ["default:"]
    rows = {}
    for v in variants:
        label, _, envs = v.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            env[k] = val
        r = subprocess.run([sys.executable, __file__, "--child"], capture_output=True, text=True, env=env, timeout=900)
        if r.returncode != 0:
            print(f"{label}: FAILED\n{r.stderr[-1500:]}")
            continue
        rows[label] = json.loads(r.stdout.strip().splitlines()[-1])
    keys = list(next(iter(rows.values())).keys()) if rows else []
    print("%-26s" % "case" + "".join("%16s" % l for l in rows))
    for k in keys:
        print("%-26s" % k + "".join("%16s" % ("%.4f ms" % rows[l][k]["ms_median"]) for l in rows))
    first = next(iter(rows.values())) if rows else {}
    for l, r in rows.items():
        bad = [k for k in keys if r[k]["checksum"] != first[k]["checksum"]]
        print(f"{l}: outputs and pass counts {'IDENTICAL to the first variant' if not bad else 'DIFFER in ' + str(bad)}")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
