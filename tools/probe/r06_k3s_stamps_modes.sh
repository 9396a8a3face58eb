set -uo pipefail
root="$(pwd)"
sed -e 's#^out=.*#out="${here}/../libvaa_hip_timing.so"#' -e "s#^here=.*#here=${root}/roboticattack_amd/csrc#" roboticattack_amd/csrc/build.sh > /tmp/build_timing.sh && bash /tmp/build_timing.sh -DVAA_K3S_TIMING > /dev/null 2>&1
for m in "" cold gemms; do echo "== mode: ${m:-warm}"; VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 300 python tools/probe/k3s_stamps.py $m 2>&1 | grep -v amdgpu; done
rm -f roboticattack_amd/libvaa_hip_timing.so
