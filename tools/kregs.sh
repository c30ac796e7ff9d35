#!/bin/bash
# register / LDS / scratch use of every kernel of a .hip source (device-only assembly, gfx950): usage tools/kregs.sh gsearch_amd/csrc/gs_join.hip [name filter] [extra hipcc flags]
S=$(mktemp --suffix=.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S -o $S $3 "$1" 2>/dev/null
python3 - "$S" "$2" <<'P'
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt and flt not in name: continue
    print("%-84s vgpr %3s sgpr %3s spill_v %3s lds %6s scratch %4s" % (name[:84], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
P
[ -n "$KEEP_ASM" ] && cp $S "$KEEP_ASM"; rm -f $S
