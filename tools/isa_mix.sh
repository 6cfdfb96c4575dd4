#!/bin/bash
# Instruction mix of the innermost (T-step) loop of one kernel in an AMDGPU assembly listing.
#   tools/isa_mix.sh file.s [mangled-name-substring] [rows]      (hipcc -S --cuda-device-only ... -o file.s)
f=$1; pat=${2:-Li4ELi1ELi1ELb0ELb0ELb0E}
k0=$(grep -n "^_ZN3mcp.*${pat}.*:" $f | head -1 | cut -d: -f1)
k1=$(awk -v s=$k0 'NR>s && /^\s*s_endpgm/{print NR; exit}' $f)
end=$(awk -v s=$k0 -v e=$k1 'NR>s && NR<e && /s_cbranch_scc0 .LBB/{print NR}' $f | tail -1)
lbl=$(sed -n ${end}p $f | awk '{print $2}')
start=$(grep -n "^${lbl}:" $f | head -1 | cut -d: -f1)
body() { sed -n ${start},${end}p $f; }
echo "$(sed -n ${k0}p $f | cut -d: -f1 | cut -c1-60) loop ${lbl}: VALU $(body | grep -c '^\s*v_')  LDS $(body | grep -c '^\s*ds_')  SMEM $(body | grep -c '^\s*s_load')  s_nop $(body | grep -c 's_nop')"
body | grep -v "^\s*;" | awk '{print $1}' | grep "^[vds]_" | sort | uniq -c | sort -rn | head -${3:-14} | tr '\n' ';'; echo
awk -v s=$k1 'NR>s && /NumVgprs|; Occupancy/{print; n++} n>=2{exit}' $f | tr '\n' ' '; echo
