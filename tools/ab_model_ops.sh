#!/usr/bin/env bash
# Same-box A/B of the model-side elementwise operators added in round 5 (wave-per-row LayerNorm, LayerScale + residual; a fused GELU pair was measured
# here too — 464.9 against 464.5 ms per step without it — and removed):
# the ViT towers run on two streams, so per-kernel times of their kernels overlap and only the un-profiled step time tells.  A B B A ...
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "${root}"
out="${root}/gpurun_out/ab"; mkdir -p "${out}"
common="--steps ${STEPS:-30} --warmup 5 --bs ${BS:-64} --no-cpu-baseline --no-kernel-suite --no-per-rank --no-configs --profile-steps 0"
i=0
for cfg in ${CFGS:-new old old new ln_only scale_only new old}; do
  i=$((i + 1))
  case "${cfg}" in
    new) envs="";;
    old) envs="VAA_LN_WAVE=0 VAA_MODEL_SCALE_ADD=0";;
    ln_only) envs="VAA_MODEL_SCALE_ADD=0";;
    scale_only) envs="VAA_LN_WAVE=0";;
  esac
  env ${envs} python bench.py ${common} --full-out "gpurun_out/ab/${i}_${cfg}_full.json" > "${out}/${i}_${cfg}.json" 2> "${out}/${i}_${cfg}.err"
  python -c "import json;d=json.load(open('${out}/${i}_${cfg}.json'));print('${i} ${cfg}', d['ms_per_step'], d['value'])"
done
