#!/bin/bash
# registers, spills, scratch and occupancy of every kernel of the library, as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage); usage: bash scripts/resource_table.sh > profiles/r05/kernel_resources.txt
cd "$(dirname "$0")/.."
printf "%-14s %-44s %6s %6s %7s %8s %9s %4s\n" file kernel VGPRs AGPRs "VGPR-sp" "SGPR-sp" "scratch-B" occ
for f in hevcbitstream_amd/csrc/*.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$f" 2>&1 |
  awk -v file="$(basename $f .hip)" '
    /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name) }
    / VGPRs: /      { v=$0; sub(/.* VGPRs: /,"",v); sub(/ .*/,"",v) }
    / AGPRs: /      { a=$0; sub(/.* AGPRs: /,"",a); sub(/ .*/,"",a) }
    /ScratchSize/   { s=$0; sub(/.*lane\]: /,"",s); sub(/ .*/,"",s) }
    /Occupancy/     { o=$0; sub(/.*SIMD\]: /,"",o); sub(/ .*/,"",o) }
    /SGPRs Spill:/  { ss=$0; sub(/.*Spill: /,"",ss); sub(/ .*/,"",ss) }
    /VGPRs Spill:/  { vs=$0; sub(/.*Spill: /,"",vs); sub(/ .*/,"",vs) }
    /LDS Size/      { cmd="echo " name " | c++filt"; cmd | getline dn; close(cmd); sub(/\(.*/,"",dn); sub(/^void /,"",dn);
                      printf "%-14s %-44s %6s %6s %7s %8s %9s %4s\n", file, substr(dn,1,44), v, a, vs, ss, s, o }'
done
