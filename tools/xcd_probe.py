"""Probe of the XCD-local phase barrier (csrc/k_xcd.h): placement of a 256-workgroup grid over the XCDs, correctness of the L2-only hand-off
(plain stores + vmcnt drain + one L2 atomic per workgroup + relaxed poll + L1-bypassing reads, no cache maintenance), cost per barrier.
Usage: python tools/xcd_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from news_recommendation_amd import _capi
dev = torch.device('cuda:0'); lib = _capi.load(); st = lambda: torch.cuda.current_stream().cuda_stream
sync = torch.zeros(32, dtype=torch.int32, device=dev)
rec = torch.zeros(8 * 32 * 512, dtype=torch.int32, device=dev)
out = torch.zeros(768, dtype=torch.int32, device=dev)
for phases in (1, 10, 100, 400):
    out.zero_()
    _capi.check(lib, lib.nr_debug_xcd_probe(sync.data_ptr(), rec.data_ptr(), out.data_ptr(), phases, st()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out.zero_()
    e0.record()
    _capi.check(lib, lib.nr_debug_xcd_probe(sync.data_ptr(), rec.data_ptr(), out.data_ptr(), phases, st()))
    e1.record(); torch.cuda.synchronize()
    o = out.cpu().numpy(); s = sync.cpu().numpy()
    xcc = o[256:512]; slot = o[512:768]
    print(f"phases {phases:4d}: {e0.elapsed_time(e1) * 1e3:9.1f} us total, {e0.elapsed_time(e1) * 1e3 / (2 * phases):6.2f} us per barrier (+ 64 KB read-back per second barrier) | stale words {int(o[:256].sum())} | "
          f"error word {int(s[16])} | workgroups per XCC {np.bincount(xcc, minlength=8).tolist()} | block % 8 == xcc: {int((xcc == np.arange(256) % 8).sum())}/256 | slots ok {sorted(slot[xcc == 0].tolist()) == list(range(32))}")
