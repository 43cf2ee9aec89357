#!/bin/bash
# tools/asm_census.sh <name> [hipcc flags]: scratch accesses per phase (between the DH_ASM_MARKERS comments) and register counts of the DMR / YSF / NXDN chain kernels.
# Run it before and after touching a rare path: one more live value there can move spills into the hot loop (a 6.8 -> 11.9 ms regression looked exactly like this).
name=$1; shift
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-parentheses-equality -DDH_ASM_MARKERS "$@" --cuda-device-only -S digiham_amd/csrc/engine.hip -o /tmp/c_$name.s 2>/dev/null
for k in "_ZN12_GLOBAL__N_17k_chainILi80ELb0ELi1ELi10ELi0EE" "_ZN12_GLOBAL__N_17k_chainILi80ELb0ELi2ELi10ELi0EE" "_ZN12_GLOBAL__N_17k_chainILi160ELb0ELi3ELi20ELi0EE"; do
a=$(grep -n "^$k.*:" /tmp/c_$name.s | cut -d: -f1)
awk -v a=$a 'NR>=a' /tmp/c_$name.s | awk '/s_endpgm/ {print; exit} {print}' > /tmp/c_$name.k.s
echo "$name $k: $(awk '/DH_PHASE/ {ph=$3} /scratch_/ {c[ph]++} END {for (p in c) printf "%s:%d ", p, c[p]}' /tmp/c_$name.k.s) | $(grep -m3 "ScratchSize\|NumVgprs" <(awk -v a=$a 'NR>a' /tmp/c_$name.s) | tr '\n' ' ')"
done
