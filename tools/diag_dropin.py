import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from news_recommendation_amd import ops
print('cpu threads', torch.get_num_threads())
cfg = bench.make_cfg('NRMS', 'small'); wl = bench.Workload('NRMS', cfg)
dev = torch.device('cuda:0')
model = wl.make_model().to(dev).train()
b = wl.batches(0, 1, 512, 'cpu')[0]
mb = wl.as_dataloader_batch(b)
parts = [x['title'] for x in mb['clicked_news']]
def T(fn, n=10):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
buf = torch.empty(512, 50, 20, dtype=torch.int64).pin_memory()
print('stack out pinned ms', T(lambda: torch.stack(parts, dim=1, out=buf)))
print('stack plain ms', T(lambda: torch.stack(parts, dim=1)))
print('minmax ms', T(lambda: (int(buf.min()), int(buf.max()))))
print('h2d ms', T(lambda: buf.to(dev, non_blocking=True)))
print('stage ms', T(lambda: ops.stack_to_device(parts, dev, cfg.num_words)))
print('forward dropin ms', T(lambda: wl.forward_dropin(model, mb)))
bd = wl.batches(0, 1, 512, dev)[0]
print('forward ids ms', T(lambda: wl.forward(model, bd)))
for nt in (1, 4, 8):
    torch.set_num_threads(nt)
    print(nt, 'threads: stack', T(lambda: torch.stack(parts, dim=1, out=buf)), 'minmax', T(lambda: (int(buf.min()), int(buf.max()))))

# ---- where does the host time of a training step go? ----
import cProfile, pstats, io
torch.set_num_threads(32)
opt = wl.make_optimizer(model)
crit = torch.nn.CrossEntropyLoss(); target = torch.zeros(512, dtype=torch.long, device=dev)
batches = wl.batches(0, 4, 512, dev)
def step(i):
    loss = crit(wl.forward(model, batches[i % 4]), target); loss.backward(); opt.step()
for i in range(5): step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(20): step(i)
te = time.perf_counter() - t
torch.cuda.synchronize()
print('train step: enqueue ms', te / 20 * 1e3, 'total ms', (time.perf_counter() - t) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])
