export TMPDIR=/tmp
timeout 500 python tools/soak_loss.py --seconds 360 --seed 91 2>&1 | tail -3
VAA_K3_ONE_PASS=0 timeout 200 python tools/soak_loss.py --seconds 90 --seed 91 2>&1 | tail -1
