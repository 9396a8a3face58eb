export TMPDIR=/tmp
timeout 300 python tools/gemm_overlap_probe.py 2400 2>&1 | grep "^M="
timeout 300 python tools/gemm_overlap_probe.py 1200 2>&1 | grep "^M="
timeout 300 python tools/gemm_overlap_probe.py 19200 2>&1 | grep "^M="
