export TMPDIR=/tmp
mkdir -p gpurun_out/r3
timeout 1500 python tools/soak_parity.py --seconds 1200 --seed 77 2>&1 | tail -3 > gpurun_out/r3/soak_parity2.txt
timeout 900 python tools/soak_loss.py --seconds 600 --seed 78 2>&1 | tail -3 > gpurun_out/r3/soak_loss2.txt
cat gpurun_out/r3/soak_parity2.txt gpurun_out/r3/soak_loss2.txt
