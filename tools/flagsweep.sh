#!/bin/bash
# Builds libapd_hip.so variants whose D=13 unit is compiled with extra compiler flags (container side), into build/flags/.
# usage: tools/flagsweep.sh name1 "flags1" name2 "flags2" ...
cd /root/repo/audio_pattern_discovery_amd/csrc
mkdir -p /root/repo/build/flags /tmp/t
args=("$@")
for ((i=0; i<${#args[@]}; i+=2)); do
  name=${args[i]}; flags=${args[i+1]}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $flags -Rpass-analysis=kernel-resource-usage -c dtw_sys_d13.hip -o /tmp/t/flag_$name.o 2> /tmp/t/flag_$name.log &
done
wait
for ((i=0; i<${#args[@]}; i+=2)); do
  name=${args[i]}
  grep -A6 "Function Name: _ZN3apd18dtw_fused_systolicILi13ELi9ELi16ELb1ELb1EEEvNS_11AlignLaunchE" /tmp/t/flag_$name.log | grep -E "VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste -sd' ' | sed "s/^/$name: /"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build/flags/libapd_hip_$name.so apd_api.o dtw_generic.o dtw_sys_d8.o dtw_sys_d10.o /tmp/t/flag_$name.o dtw_sys_d16.o dtw_sys_d20.o dtw_sys_d26.o dtw_wf_d8.o dtw_wf_d10.o dtw_wf_d13.o dtw_wf_d16.o dtw_wf_d20.o dtw_wf_d26.o clustering.o companions.o comm.o formats.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
done
