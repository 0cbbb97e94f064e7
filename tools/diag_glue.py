"""Which Python lines launch the torch (non-engine) kernels of a training step?  Usage: python tools/diag_glue.py [MODEL] [SHAPE]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
M = sys.argv[1] if len(sys.argv) > 1 else 'NRMS'; SH = sys.argv[2] if len(sys.argv) > 2 else 'small'
cfg = bench.make_cfg(M, SH, 0); wl = bench.Workload(M, cfg)
dev = torch.device('cuda:0')
model = wl.make_model().to(dev).train()
opt = wl.make_optimizer(model)
crit = torch.nn.CrossEntropyLoss(); target = torch.zeros(512, dtype=torch.long, device=dev)
batches = wl.batches(0, 2, 512, dev)
def step(i):
    loss = wl.loss(model, batches[i % 2], crit, target); loss.backward(); opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
NS = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(NS): step(i)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=12):
    us = getattr(e, 'self_device_time_total', 0)
    if us <= 0 or not e.key.startswith('aten::'):
        continue
    st = e.stack or []
    site = next((f for f in st if 'news_recommendation_amd' in f or 'bench.py' in f), st[0] if st else '?')
    rows.append((us / NS, e.count / NS, e.key, site.split('/root/repo/')[-1][:120]))
tot = 0
for us, n, name, site in sorted(rows, reverse=True)[:50]:
    print(f"{us:8.1f} us/step {n:5.1f}x  {name:28s} {site}")
    tot += us
print('total aten self device time per step (listed):', round(tot, 1))
