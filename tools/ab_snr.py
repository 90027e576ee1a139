#!/usr/bin/env python3
"""A/B of decoder variants over a sweep of noise levels (the pass-count mix changes with Es/N0): BASELINE configs[1],
1024 x BG1 Zc=384 R=1/3, numMaxIter 8.  Every variant runs in its own process on the same seeded inputs.
  python tools/ab_snr.py p0:NRLDPC_HIP_PERSIST=0 p1:NRLDPC_HIP_PERSIST=1"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SNRS = [float(x) for x in os.environ.get("AB_SNRS", "-0.5,0,0.25,0.5,0.75,1,1.5,2,3").split(",")]


def child():
    import numpy as np
    import torch
    import bench as B
    import openairinterface5g_amd as pkg
    pkg.LDPCinit()
    out = torch.zeros((B.BATCH, B.NUM_LLR // 8), dtype=torch.uint8, device="cuda")
    it = torch.zeros(B.BATCH, dtype=torch.int32, device="cuda")
    res = {}
    for snr in SNRS:
        _, llr = B.make_batch(pkg, torch, snr, 3000)
        for _ in range(5):
            pkg.decode_batch_device(B.BG, B.Z, B.R, llr, out, it, numMaxIter=B.MAX_ITER)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in range(40):
            pkg.decode_batch_device(B.BG, B.Z, B.R, llr, out, it, numMaxIter=B.MAX_ITER)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
        h = np.bincount(it.cpu().numpy(), minlength=11)
        res[f"{snr:+.2f} dB"] = dict(ms_median=ms[20], passes=float(it.float().mean()), hist=h.tolist(),
                                     checksum=int(out.to(torch.int64).sum()) + int(it.sum()))
    print(json.dumps(res))


def main():
    rows = {}
    for v in sys.argv[1:] or ["default:"]:
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
    first = next(iter(rows.values()))
    print("%-10s %-7s %-44s" % ("Es/N0", "passes", "blocks by pass count 3..10") + "".join("%14s" % l for l in rows))
    for k in first:
        print("%-10s %-7.2f %-44s" % (k, first[k]["passes"], str(first[k]["hist"][3:])) + "".join("%14s" % ("%.4f ms" % rows[l][k]["ms_median"]) for l in rows))
    for l, r in rows.items():
        bad = [k for k in first if r[k]["checksum"] != first[k]["checksum"]]
        print(f"{l}: outputs and pass counts {'IDENTICAL to the first variant' if not bad else 'DIFFER in ' + str(bad)}")


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
