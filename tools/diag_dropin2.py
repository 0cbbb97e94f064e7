import sys, time, torch, numpy as np, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import bench
from news_recommendation_amd import ops
M = sys.argv[1] if len(sys.argv) > 1 else 'NRMS'; SH = sys.argv[2] if len(sys.argv) > 2 else 'small'
cfg = bench.make_cfg(M, SH, 0); wl = bench.Workload(M, cfg)
dev = torch.device('cuda:0')
model = wl.make_model().to(dev).train()
opt = wl.make_optimizer(model)
crit = torch.nn.CrossEntropyLoss(); target = torch.zeros(512, dtype=torch.long, device=dev)
cpu_batches = [wl.as_dataloader_batch(b) for b in wl.batches(0, 2, 512, 'cpu')]
def step(i):
    loss = crit(wl.forward_dropin(model, cpu_batches[i % 2]), target); loss.backward(); opt.step()
for i in range(4): step(i)
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    for i in range(20): step(i)
    te = time.perf_counter() - t
    torch.cuda.synchronize()
    print('dropin train step: enqueue ms', te / 20 * 1e3, 'total ms', (time.perf_counter() - t) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(20): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:4000])
