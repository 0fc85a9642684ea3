#!/bin/bash
# kernel resource usage of libdellyhip.so (VGPRs, scratch, LDS, occupancy) from hipcc's remarks; [extra hipcc flags...]
cd "$(dirname "$0")/../delly_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value "$@" \
  -Rpass-analysis=kernel-resource-usage -o /tmp/dellyhip_ru.so dellyhip.hip 2>&1 |
  python3 -c '
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur={"name":m.group(1)}; continue
    for key,pat in (("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)")):
        m=re.search(pat,line)
        if m and cur is not None: cur[key]=int(m.group(1))
    if cur and "lds" in cur:
        print("%-70s vgpr %3d agpr %3d scratch %4d occ %d lds %6d"%(cur["name"][:70],cur.get("vgpr",0),cur.get("agpr",0),cur.get("scratch",0),cur.get("occ",0),cur["lds"])); cur=None
'
