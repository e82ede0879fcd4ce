#!/bin/bash
# Builds libapd_hip.so variants with the D=13 unit ablated (container side), into gpurun_out/ablate/.
cd /root/repo/audio_pattern_discovery_amd/csrc
mkdir -p /root/repo/build/ablate /tmp/t
for a in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DAPD_ABLATE=$a -c dtw_sys_d13.hip -o /tmp/t/abl_$a.o &
done
wait
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build/ablate/libapd_hip_abl$a.so apd_api.o dtw_generic.o dtw_sys_d8.o dtw_sys_d10.o /tmp/t/abl_$a.o dtw_sys_d16.o dtw_sys_d20.o dtw_sys_d26.o dtw_sysx_d8.o dtw_sysx_d10.o dtw_sysx_d16.o dtw_sysx_d20.o dtw_sysx_d26.o ${SYSX13:-dtw_sysx_d13.o} dtw_wf_d8.o dtw_wf_d10.o dtw_wf_d13.o dtw_wf_d16.o dtw_wf_d20.o dtw_wf_d26.o clustering.o companions.o comm.o formats.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
done
ls -la /root/repo/build/ablate/
