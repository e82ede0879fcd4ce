#!/bin/bash
# Builds libapd_hip.so variants in which ONE unit (default dtw_sys_d13; UNIT=dtw_sysx_d13 for strict mode's) is compiled with other
# compiler flags (container side), into build/flags/.  Prints the registers of KERNEL (mangled name) for each.
# usage: [UNIT=dtw_sysx_d13] [KERNEL=...] tools/flagsweep.sh name1 "flags1" name2 "flags2" ...
cd /root/repo/audio_pattern_discovery_amd/csrc
UNIT=${UNIT:-dtw_sys_d13}
if [ "$UNIT" = dtw_sysx_d13 ]; then KERNEL=${KERNEL:-_ZN3apd18dtw_fused_systolicILi13ELi9ELi16ELb1ELb0EEEvNS_11AlignLaunchE}; fi
KERNEL=${KERNEL:-_ZN3apd18dtw_fused_systolicILi13ELi9ELi16ELb1ELb1EEEvNS_11AlignLaunchE}
mkdir -p /root/repo/build/flags /tmp/t
OBJS=$(make -pn 2>/dev/null | sed -n 's/^OBJS := //p' | head -1)
[ -n "$OBJS" ] || { echo "cannot read OBJS from the Makefile"; exit 1; }
args=("$@")
for ((i=0; i<${#args[@]}; i+=2)); do
  name=${args[i]}; flags=${args[i+1]}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $flags -Rpass-analysis=kernel-resource-usage -c $UNIT.hip -o /tmp/t/flag_$name.o 2> /tmp/t/flag_$name.log &
done
wait
for ((i=0; i<${#args[@]}; i+=2)); do
  name=${args[i]}
  grep -A6 "Function Name: $KERNEL" /tmp/t/flag_$name.log | grep -E "VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste -sd' ' | sed "s/^/$name: /"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build/flags/libapd_hip_$name.so ${OBJS//$UNIT.o//tmp/t/flag_$name.o} -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
done
