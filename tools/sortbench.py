import torch, time
dev='cuda:0'
g=torch.Generator().manual_seed(0)
for n in (542720, 1899520):
    ids=torch.randint(0,70976,(n,),generator=g).to(dev)
    for name,fn in [('int64', lambda: torch.sort(ids)), ('int32', lambda: torch.sort(ids.to(torch.int32))), ('int64 stable', lambda: torch.sort(ids, stable=True)),
                    ('argsort int32', lambda: torch.argsort(ids.to(torch.int32))), ('int16hi', lambda: torch.sort((ids).to(torch.int32) )),]:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0=time.perf_counter()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record()
        t1=time.perf_counter()
        torch.cuda.synchronize()
        print(n, name, 'gpu_us', round(e0.elapsed_time(e1)*100,1), 'host_us', round((t1-t0)*1e5,1))
