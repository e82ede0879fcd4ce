#!/bin/bash
# Lists VGPR / scratch / occupancy of every alignment kernel instantiation; exits 1 if one of them uses scratch.
cd "$(dirname "$0")/../audio_pattern_discovery_amd/csrc"
for fam in sys wf; do for d in 8 10 13 16 20 26; do
  extra=""; [ "$fam" = "sys" ] && extra="-mllvm -amdgpu-sched-strategy=iterative-ilp"; [ "$fam$d" = "sys13" ] && extra="$extra -mllvm -misched-cluster=false"       # as the Makefile builds it
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $extra --cuda-device-only -Rpass-analysis=kernel-resource-usage -c dtw_${fam}_d$d.hip -o /dev/null 2>&1 \
   | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - \
   | sed -E 's/Function Name: _ZN3apd18dtw_fused_systolicILi([0-9]+)ELi([0-9]+)ELi([0-9]+)ELb([01])ELb([01])EEEvNS_11AlignLaunchE/systolic D=\1 C=\2 G=\3 uniform=\4 hybrid=\5/' > /tmp/apd_res_${fam}_$d.txt &
done; done
wait
cat /tmp/apd_res_*_*.txt
if grep -v "ScratchSize \[bytes/lane\]: 0" /tmp/apd_res_*_*.txt | grep -q Scratch; then echo "SCRATCH IN USE"; exit 1; fi
echo "no scratch"
