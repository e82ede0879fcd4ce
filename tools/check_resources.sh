#!/bin/bash
# Lists VGPR / scratch / occupancy of every alignment kernel instantiation; exits 1 if one of them uses scratch.
cd "$(dirname "$0")/../audio_pattern_discovery_amd/csrc"
for fam in sys sysx wf; do for d in 8 10 13 16 20 26; do
  extra=""; [ "$fam" = "sys" ] && extra="-mllvm -amdgpu-sched-strategy=iterative-ilp"; [ "$fam$d" = "sys13" ] && extra="$extra -mllvm -misched-cluster=false"       # as the Makefile builds it
  [ "$fam" = "sysx" ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $extra --cuda-device-only -Rpass-analysis=kernel-resource-usage -c dtw_${fam}_d$d.hip -o /dev/null 2>&1 \
   | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - \
   | sed -E 's/Function Name: _ZN3apd18dtw_fused_systolicILi([0-9]+)ELi([0-9]+)ELi([0-9]+)ELb([01])ELb([01])EEEvNS_11AlignLaunchE/systolic D=\1 C=\2 G=\3 uniform=\4 hybrid=\5/' > /tmp/apd_res_${fam}_$d.txt &
done; done
# the UPGMA / companion kernels too: a launch that needs scratch pays for it every time, and the UPGMA loop is made of short launches
# (round 4: two debug stamps in the commit pushed upgma_segment_kernel to 400 bytes of scratch per lane and doubled every dendrogram)
for unit in clustering companions; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $unit.hip -o /dev/null 2>&1 \
   | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - > /tmp/apd_res_${unit}_0.txt &
done
wait
cat /tmp/apd_res_*_*.txt
if grep -v "ScratchSize \[bytes/lane\]: 0" /tmp/apd_res_*_*.txt | grep -q Scratch; then echo "SCRATCH IN USE"; exit 1; fi
echo "no scratch"
