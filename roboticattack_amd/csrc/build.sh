#!/usr/bin/env bash
# Builds libvaa_hip.so for gfx950 (MI355X) in-tree. -ffp-contract=off: FMAs are explicit (parity contract).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libvaa_hip.so"
tmp="${out}.tmp.$$"  # compile to a private name, then rename: a concurrent loader never sees a half-written library
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function \
  "${here}/vaa_capi.hip" "${here}/vaa_patch_fwd.hip" "${here}/vaa_patch_grad.hip" "${here}/vaa_patch_resize.hip" "${here}/vaa_loss.hip" "${here}/vaa_head.hip" "${here}/vaa_head_slice.hip" "${here}/vaa_update.hip" "${here}/vaa_patch_eval.hip" "${here}/vaa_model_ops.hip" "${here}/vaa_attention.hip" \
  -o "${tmp}" "$@"
mv -f "${tmp}" "${out}"
echo "built ${out}"
