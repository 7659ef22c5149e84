#!/usr/bin/env python3
"""Compact kernel resource table: tools/kres.py [unit=mcf_kernels] [filter] — VGPRs / AGPRs / scratch / occupancy / LDS per kernel
from hipcc -Rpass-analysis=kernel-resource-usage (the Makefile's flags)."""
import re, subprocess, sys
unit = sys.argv[1] if len(sys.argv) > 1 else "mcf_kernels"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ("/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -disable-machine-licm -Wno-unused-function "
       "-Wno-unused-value -Wno-pass-failed -Rpass-analysis=kernel-resource-usage -c -o /dev/null microclimf_amd/csrc/%s.hip" % unit)
out = subprocess.run(cmd.split(), capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln) or re.search(r" Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("mcf::", "").replace("(SolveArgs)", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w \[\]/]+): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-58s %5s %5s %6s %7s %4s %7s" % ("kernel", "VGPR", "SGPR", "sspill", "scratch", "occ", "LDS"))
for k, r in rows.items():
    if flt and flt not in k:
        continue
    print("%-58s %5s %5s %6s %7s %4s %7s" % (k[:58], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
                                          r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
