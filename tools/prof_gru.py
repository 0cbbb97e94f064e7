"""Run the GRU forward/backward step kernels in a loop (for rocprofv3 --pmc passes / timing).  Usage: python tools/prof_gru.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from news_recommendation_amd import ops_gru
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = 'cuda:0'
torch.manual_seed(0)
gru = torch.nn.GRU(900, 900).to(dev)
x = (torch.randn(B, 50, 900, device=dev) * 0.5).requires_grad_(True)
h0 = (torch.randn(B, 900, device=dev) * 0.5).requires_grad_(True)
lens = torch.full((B,), 50, dtype=torch.long)
for it in range(3):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    out = ops_gru.gru_last_state(x, h0, lens, gru)
    e1.record()
    out.sum().backward()
    e2.record()
    torch.cuda.synchronize()
    print('fwd %.2f ms  bwd %.2f ms' % (e0.elapsed_time(e1), e1.elapsed_time(e2)))
