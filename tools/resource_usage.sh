#!/bin/bash
# Register / scratch / LDS use of every kernel of libgsmcal.so (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
# "No kernel of the latency path carries a scratch frame"  is checked with this.
cd "$(dirname "$0")/../multi-rtl-sdr-calibration_amd" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC csrc/gsmcal.hip -o /tmp/gsmcal_ru.so -ldl \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys, re, subprocess
name = None
rec = {}
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m:
        name = m.group(1); rec[name] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?:\s+(\d+)', ln)
    if m and name:
        rec[name][m.group(1).strip()] = m.group(2)
for n, r in rec.items():
    try:
        d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    except Exception:
        d = n
    print(d[:64].ljust(64), 'VGPR', r.get('VGPRs', '?').rjust(3), 'AGPR', r.get('AGPRs', '?').rjust(3), 'SGPR', r.get('TotalSGPRs', '?').rjust(3),
          'scratch', r.get('ScratchSize', '?').rjust(4), 'occ', r.get('Occupancy', '?'), 'LDS', r.get('LDS Size', '?'))
"
