export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model_ops.py -x -q 2>&1 | grep "passed\|failed\|Error" | head
for bs in 8 4 16; do
for q in 0 1; do
VAA_FUSED_QKV=$q timeout 600 python bench.py --bs $bs --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-suite --no-per-rank > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
python -c "import json; d=json.load(open('/tmp/o.json')); print('QKV bs $bs fused $q: ms/step %.2f  img/s %.1f  finite %s' % (d['ms_per_step'], $bs*1000/d['ms_per_step'], d['loss_finite']))"
done; done
