#!/bin/bash
# Register / scratch usage of the group and aggregation kernels (cross-compile, no GPU needed).
cd "$(dirname "$0")/../lfbm5d_amd/csrc"
f=${1:-lfbm5d_group_ht.hip}
extra=""; [ "$f" = lfbm5d_bm.hip -o "$f" = lfbm5d_scan2.hip ] && extra="-ffp-contract=off"
hipcc -O3 --offload-arch=gfx950 -std=c++17 $extra -S --cuda-device-only -o /tmp/kregs.s $f 2>&1 | grep -E "error" -A3
python3 - <<'PY'
import re
t=open('/tmp/kregs.s').read()
for b in t.split('  - .agpr_count')[1:]:
    g=lambda k: re.search(r'\.%s:\s+(\S+)'%k,b).group(1)
    print('%-60s vgpr %3s sgpr %3s spill %s scratch %s lds %s'%(g('name')[22:80],g('vgpr_count'),g('sgpr_count'),g('vgpr_spill_count'),g('private_segment_fixed_size'),g('group_segment_fixed_size')))
PY
