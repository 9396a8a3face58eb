export TMPDIR=/tmp
timeout 300 python tools/k3_onepass_check.py 2>&1 | tail -12
