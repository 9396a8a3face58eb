#!/usr/bin/env bash
# K3s quick loop: parity tests, per-dispatch / in-stream / cold times, phase stamps (8- and 16-column forms)
set -uo pipefail
root="$(pwd)"; out="${root}/gpurun_out/k3s_quick"; mkdir -p "${out}"
mk() { sed -e "s#^out=.*#out=\"\${here}/../$1\"#" -e "s#^here=.*#here=${root}/roboticattack_amd/csrc#" roboticattack_amd/csrc/build.sh > /tmp/build_var.sh && bash /tmp/build_var.sh "${@:2}" > /dev/null 2>&1; }
{ timeout 900 python -m pytest tests/test_gpu_head_slice.py -x -q 2>&1 | tail -4; VAA_K3S_COLS=16 timeout 900 python -m pytest tests/test_gpu_head_slice.py -x -q 2>&1 | tail -4; } > "${out}/tests.txt" 2>&1
{ echo "== 8 columns (default)"; timeout 300 python tools/k3s_bench.py 128 16; echo "== 16 columns"; VAA_K3S_COLS=16 timeout 300 python tools/k3s_bench.py 128 16; } > "${out}/k3s_bench.txt" 2>&1
mk libvaa_hip_timing.so -DVAA_K3S_TIMING
{ for c in 8 16; do echo "== ${c} columns"; VAA_K3S_COLS=${c} VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py; VAA_K3S_COLS=${c} VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py cold; done; } > "${out}/k3s_stamps.txt" 2>&1
rm -f roboticattack_amd/libvaa_hip_timing.so
cat "${out}/tests.txt"; grep "==\|phases alone\|in a stream\|COLD\|per dispatch" "${out}/k3s_bench.txt" | cut -c1-330; grep -v amdgpu.ids "${out}/k3s_stamps.txt" | grep "==\|R=\|ph1 loop\|first poll\|stats done\|  end"
