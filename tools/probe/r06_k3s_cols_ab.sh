#!/usr/bin/env bash
# K3s: 8 action columns per workgroup (32 workgroups per row block, the default) against 16 (VAA_K3S_COLS=16), and the B operand in front of the
# first poll (default in the 8-column form) against in quarters between the stages of the statistics (-DVAA_K3S_B_EARLY=0); parity tests first.
set -uo pipefail
root="$(pwd)"; out="${root}/gpurun_out/k3s_cols"; mkdir -p "${out}"
mk() { sed -e "s#^out=.*#out=\"\${here}/../$1\"#" -e "s#^here=.*#here=${root}/roboticattack_amd/csrc#" roboticattack_amd/csrc/build.sh > /tmp/build_var.sh && bash /tmp/build_var.sh "${@:2}" > /dev/null 2>&1; }
{ timeout 900 python -m pytest tests/test_gpu_head_slice.py -x -q 2>&1 | tail -4; VAA_K3S_COLS=16 timeout 900 python -m pytest tests/test_gpu_head_slice.py -x -q 2>&1 | tail -4; } > "${out}/tests.txt" 2>&1
{ echo "== 8 columns (default)"; timeout 300 python tools/k3s_bench.py 128 16; echo "== 16 columns"; VAA_K3S_COLS=16 timeout 300 python tools/k3s_bench.py 128 16; } > "${out}/k3s_bench.txt" 2>&1
mk libvaa_hip_late.so -DVAA_K3S_B_EARLY=0
{ echo "== 8 columns, B in quarters between the stages (-DVAA_K3S_B_EARLY=0)"; VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_late.so" timeout 300 python tools/k3s_bench.py 128 16; } >> "${out}/k3s_bench.txt" 2>&1
mk libvaa_hip_ring8.so -DVAA_SLICE_RING8=8
{ echo "== 8 columns, ring of 8 groups (96 KB: one workgroup per CU -> two launches by the residency rule)"; VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_ring8.so" timeout 300 python tools/k3s_bench.py 128 16; } >> "${out}/k3s_bench.txt" 2>&1
mk libvaa_hip_timing.so -DVAA_K3S_TIMING
{ for c in 8 16; do echo "== ${c} columns"; VAA_K3S_COLS=${c} VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py; VAA_K3S_COLS=${c} VAA_LIB_PATH="${root}/roboticattack_amd/libvaa_hip_timing.so" timeout 200 python tools/probe/k3s_stamps.py cold; done; } > "${out}/k3s_stamps.txt" 2>&1
rm -f roboticattack_amd/libvaa_hip_late.so roboticattack_amd/libvaa_hip_ring8.so roboticattack_amd/libvaa_hip_timing.so
cat "${out}/tests.txt"; grep -v "^ *$" "${out}/k3s_bench.txt" | cut -c1-260
