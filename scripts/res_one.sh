#!/bin/bash
# registers / spills / scratch / occupancy of the kernels of ONE .hip file (dev aid): bash scripts/res_one.sh file.hip [-Dflags...]
f="$1"; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ihevcbitstream_amd/csrc -Wall -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$f" 2>&1 |
  awk '
    /error/ { print }
    /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name) }
    / VGPRs: /      { v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v) }
    /TotalSGPRs: /      { sg=$0; sub(/.*TotalSGPRs: /,"",sg); sub(/ .*/,"",sg) }
    /ScratchSize/   { s=$0; sub(/.*lane\]: /,"",s); sub(/ .*/,"",s) }
    /Occupancy/     { o=$0; sub(/.*SIMD\]: /,"",o); sub(/ .*/,"",o) }
    /SGPRs Spill:/  { ss=$0; sub(/.*Spill: /,"",ss); sub(/ .*/,"",ss) }
    /VGPRs Spill:/  { vs=$0; sub(/.*Spill: /,"",vs); sub(/ .*/,"",vs) }
    /LDS Size/      { lds=$0; sub(/.*block\]: /,"",lds); sub(/ .*/,"",lds); cmd="echo " name " | c++filt"; cmd | getline dn; close(cmd); sub(/\(.*/,"",dn); sub(/^void /,"",dn);
                      printf "%-48s VGPR %4s SGPR %4s vsp %3s ssp %4s scratch %4s occ %2s lds %6s\n", substr(dn,1,48), v, sg, vs, ss, s, o, lds }'
