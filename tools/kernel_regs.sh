#!/bin/bash
# VGPR / spill / scratch / LDS of every kernel of the built objects (no GPU needed):   bash tools/kernel_regs.sh [objects...]
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
for F in ${@:-openairinterface5g_amd/csrc/build/*.o}; do
  $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$F" "$T/fat" 2>/dev/null || continue
  [ -s "$T/fat" ] || continue
  for tgt in $($LLVM/clang-offload-bundler --type=o --input="$T/fat" --list | grep gfx950); do
    $LLVM/clang-offload-bundler --type=o --input="$T/fat" --targets="$tgt" --output="$T/co" --unbundle
    $LLVM/llvm-readelf --notes "$T/co" | python3 -c '
import sys, re
rec = {}
def flush():
    if rec.get("name"):
        print("%4s VGPR %3s spill %5s B scratch %7s B LDS(static)  %s" % (rec.get(".vgpr_count"), rec.get(".vgpr_spill_count"), rec.get(".private_segment_fixed_size"), rec.get(".group_segment_fixed_size"), rec["name"]))
for line in sys.stdin:
    m = re.search(r"\.name:\s+(\S+)", line)
    if m:
        rec["name"] = m.group(1)
    for k in (".vgpr_count", ".group_segment_fixed_size", ".vgpr_spill_count", ".private_segment_fixed_size"):
        m = re.search(re.escape(k) + r":\s+(\d+)", line)
        if m: rec[k] = m.group(1)
    if ".wavefront_size" in line:
        flush(); rec = {}
' | c++filt
  done
done
rm -rf "$T"
