#!/usr/bin/env python3
"""Timeline of the LAST nrLDPC_hip_ulsch_decode call in a rocprofv3 --kernel-trace --memory-copy-trace run of
tools/host_timeline.py: every copy and kernel with start / end relative to the call's first event (us)."""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0][:40]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "copy"))[:40]))
ev.sort()
# the last call: events after the last gap of more than 300 us that precedes a tb_rx kernel
idx = [i for i, e in enumerate(ev) if "tb_rx" in e[2]]
last = idx[-1]
start = last
while start > 0 and ev[start][0] - ev[start - 1][1] < 300_000:
    start -= 1
t0 = ev[start][0]
for s, e, n in ev[start:last + 4]:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  {n}")
