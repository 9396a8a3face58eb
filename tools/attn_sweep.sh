#!/usr/bin/env bash
# Model-side attention kernels: row groups per wave (VAA_ATTN_G = forward / dq / dk-dv digits) swept over the three shapes of the bs=64 step,
# numerics check included, then the per-kernel split under rocprofv3 for two settings and one counter pass.
#   gpurun --timeout 1500 -- 'bash tools/attn_sweep.sh'          -> gpurun_out/attn/
set -uo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="${root}/gpurun_out/attn"; mkdir -p "${out}"; export TMPDIR=/tmp; cd "${root}"
# ATTN_CFGS: "G:PF" pairs (row groups per wave : tiles requested ahead [VAA_ATTN_PF: only in the experiment build of commit 0dd8d0c]; three digits each = forward / dq / dk-dv; "-" = default)
for cfg in ${ATTN_CFGS:-111:111 211:111 121:111 112:111 222:111}; do
  g="${cfg%%:*}"; pf="${cfg##*:}"; envs=""
  [ "${g}" != "-" ] && envs="${envs} VAA_ATTN_G=${g}"; [ "${pf}" != "-" ] && envs="${envs} VAA_ATTN_PF=${pf}"
  echo "== ${envs:- defaults}"
  env ${envs} timeout 300 python tools/attn_bench.py --check --iters 20 2>&1 | grep -E '"check"|"shape"' | cut -c1-175
done | tee "${out}/sweep.txt"
for cfg in ${ATTN_PROF-111:111 222:111}; do
  g="${cfg%%:*}"; pf="${cfg##*:}"; envs=""; tag="${g}_${pf}"
  [ "${g}" != "-" ] && envs="${envs} VAA_ATTN_G=${g}"; [ "${pf}" != "-" ] && envs="${envs} VAA_ATTN_PF=${pf}"
  env ${envs} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_attn_${tag} -o a -- python tools/attn_bench.py --iters 10 > /dev/null 2> "${out}/prof_${tag}.err"
  f="$(find /tmp/prof_attn_${tag} -name '*kernel_stats.csv' | head -1)"; cp "${f}" "${out}/attn_${tag}_kernel_stats.csv" 2>/dev/null
  echo "== per kernel,${envs:- defaults}"; python - "${out}/attn_${tag}_kernel_stats.csv" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_' in r['Name']: print('  %-62s calls %4s avg %8.1f us' % (r['Name'].split('(')[0][-62:], r['Calls'], float(r['AverageNs'])/1e3))
P
done | tee "${out}/per_kernel.txt"
if [ -n "${ATTN_PMC-1}" ]; then
  for shape in llm; do
    for ctrs in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
      echo "== pmc ${shape}: ${ctrs}"; timeout 300 python tools/pmc_attn.py ${shape} ${ctrs} 2>&1 | tail -6
    done
  done | tee "${out}/pmc.txt"
fi
