#!/usr/bin/env bash
# Builds an instrumented / ablated variant of libvaa_hip.so next to this script (never the product library):
#   tools/scratch/build_variant.sh K2TIMING -DVAA_K2_TIMING        -> tools/scratch/libvaa_K2TIMING.so   (k2timing.py)
#   tools/scratch/build_variant.sh K1TIMING -DVAA_K1_TIMING        -> tools/scratch/libvaa_K1TIMING.so   (k1timing.py)
#   tools/scratch/build_variant.sh NO_ATOMICS -DVAA_K2_ABLATE_NO_ATOMICS ;  ... NO_LOADS -DVAA_K2_ABLATE_NO_LOADS   (VAA_LIB_PATH=... tools/k2exp.py)
# Tuning knobs the sources read (defaults = the product's values): -DVAA_K2_TARGET_WGS=128 (workgroups the one-channel K2 paths aim for when
# choosing row bands), -DVAA_EMBED_GROUP=2 (k-chunks per weight request group of the K2' tile kernel), -DVAA_EMBED_WAVES=8 (its waves per
# workgroup: two column blocks each), -DVAA_RESIZE_WGS=8192 (workgroups from which the resize adjoint walks several images per workgroup).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
src="${here}/../../roboticattack_amd/csrc"
name="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -I"${src}" "$@" \
  "${src}/vaa_capi.hip" "${src}/vaa_patch_fwd.hip" "${src}/vaa_patch_grad.hip" "${src}/vaa_patch_resize.hip" "${src}/vaa_loss.hip" "${src}/vaa_head.hip" \
  "${src}/vaa_update.hip" "${src}/vaa_patch_eval.hip" "${src}/vaa_model_ops.hip" "${src}/vaa_attention.hip" -o "${here}/libvaa_${name}.so"
echo "built ${here}/libvaa_${name}.so"
