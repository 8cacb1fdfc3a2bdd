#!/bin/bash
# Register / scratch use of the rendering kernels: bash tools/kernel_regs.sh [extra compiler flags]
cd "$(dirname "$0")/../bidirectional-pathtracing_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics "$@" --cuda-device-only -c render.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for n, r in rows.items():
    if "ILb1" in n: continue
    print("%-60s vgpr %4s scratch %5s B  spilled %3s  waves/SIMD %s  lds %6s" % (n[:60], r.get("VGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
