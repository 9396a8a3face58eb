"""torch.profiler view of one bench-shaped step: which framework ops launch the remaining copy/elementwise kernels."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roboticattack_amd import synthetic
from roboticattack_amd.labels import mask_labels
from roboticattack_amd.openvla_model import build_openvla
from torch.profiler import profile, ProfilerActivity
dev = "cuda"
m = build_openvla(device=dev)
B = 64
ids, labels, _ = synthetic.synth_text_batch(0, B, 18, 24)  # same call as bench.py?
labels = mask_labels(labels, [0]).to(dev)
ids = ids.to(dev)
rows_idx = m.label_row_index(labels)
pix0 = torch.randn(B, 6, 224, 224, device=dev).to(torch.bfloat16)
def step():
    pix = pix0.clone().requires_grad_(True)
    z = m.forward_rows(ids, pix, labels, rows_idx)
    z.float().square().mean().backward()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=400, max_name_column_width=60, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=50))
