"""Which Python lines issue the torch (non-engine) device operations of a training step?  A TorchDispatchMode records every aten call that
produces or writes a device tensor, with the innermost frame inside this repository.  Usage: python tools/diag_glue2.py [MODEL] [SHAPE]"""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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
for i in range(3): step(i)
torch.cuda.synchronize()
VIEWS = {'view', 'reshape', 'slice', 'select', 'expand', 'as_strided', 'unsqueeze', 'squeeze', 'detach', 'alias', 't', 'transpose', 'permute',
         'unbind', 'split', 'narrow', '_unsafe_view', 'size', 'stride', 'numel', 'dim', 'lift_fresh', 'chunk', 'unflatten', 'flatten', 'split_with_sizes',
         '_reshape_alias', 'unsafe_split', 'diagonal', 'movedim'}
log = collections.Counter(); nbytes = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        base = name.split('.')[1] if '.' in name else name
        if base in VIEWS or base.startswith(('sym_', 'is_', 'empty')):
            return out
        ts = [t for t in (out if isinstance(out, (tuple, list)) else [out]) if isinstance(t, torch.Tensor)]
        if not ts or not any(t.is_cuda for t in ts):
            return out
        st = traceback.extract_stack()
        site = next((f"{f.filename.split('/root/repo/')[-1]}:{f.lineno} {f.name}" for f in reversed(st)
                     if ('news_recommendation_amd' in f.filename or 'bench.py' in f.filename or 'diag_glue2' in f.filename) and '__torch_dispatch__' != f.name), '?')
        log[(name, site)] += 1; nbytes[(name, site)] += sum(t.numel() * t.element_size() for t in ts)
        return out
with Mode():
    step(0)
torch.cuda.synchronize()
for (name, site), n in sorted(log.items(), key=lambda kv: -nbytes[kv[0]]):
    print(f"{n:3d}x {nbytes[(name, site)] / 1e6:9.3f} MB  {name:36s} {site}")
