"""A/B the mhsa forward variants inside one process (interleaved rounds). Usage: python tools/variants.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ['x', 'mhsa_infer']
import importlib.util
spec = importlib.util.spec_from_file_location('pk', os.path.join(os.path.dirname(__file__), 'prof_kernel.py'))
pk = importlib.util.module_from_spec(spec); spec.loader.exec_module(pk)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(3):
    for v in ('2', '81'):
        os.environ['NR_MHSA_VARIANT'] = v
        print(rnd, v, 'infer %.1f us' % timeit(pk.fns['mhsa_infer']), 'train %.1f us' % timeit(pk.fns['mhsa_train']), flush=True)
