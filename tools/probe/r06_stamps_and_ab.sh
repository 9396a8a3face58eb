set -uo pipefail
root="$(pwd)"; out="${root}/gpurun_out/measure"; mkdir -p "${out}"
sed -e 's#^out=.*#out="${here}/../libvaa_hip_timing.so"#' -e "s#^here=.*#here=${root}/roboticattack_amd/csrc#" roboticattack_amd/csrc/build.sh > /tmp/build_timing.sh && bash /tmp/build_timing.sh -DVAA_K3S_TIMING > /dev/null 2>&1
{ VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py; VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py cold; } > "${out}/k3s_stamps.txt" 2>&1
rm -f roboticattack_amd/libvaa_hip_timing.so
bash tools/measure_configs.sh k3s cfg > gpurun_out/measure_configs.log 2>&1
tail -5 gpurun_out/measure_configs.log
