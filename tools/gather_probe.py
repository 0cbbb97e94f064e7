"""The north_star's stand-alone embedding-gather probe under rocprofv3: nr_gather_rows_f32 at bench.py's two points.
Usage: python tools/gather_probe.py {workload|hbm} [reps]
  workload  the B = 512 batch's 542,720 title tokens (Zipf ids, ~45 % padding id 0) on the 70,976-row table (85 MB: Infinity-Cache resident)
  hbm       542,720 uniform ids over a 400,001-row table (480 MB > the 256 MB Infinity Cache: rows come from HBM)
Run under `rocprofv3 --kernel-trace --stats` (durations) and under separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (tools/gather_prof.sh)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from news_recommendation_amd import _capi, synth

point = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
lib = _capi.load()
B = 512
if point == 'workload':
    V = 70976
    news = synth.news_titles(np.random.default_rng(0), 65238, 20, V)
    cand, hist = synth.train_batch(np.random.default_rng(1000), news, B)
    c, h = synth.batch_token_ids(news, cand, hist)
    ids = torch.from_numpy(np.concatenate([c.reshape(-1), h.reshape(-1)])).to(dev)
else:
    V = 400001
    ids = torch.randint(1, V, (B * 53 * 20,), device=dev)
table = torch.randn(V, 300, device=dev)
out = torch.empty(ids.numel(), 300, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    _capi.check(lib, lib.nr_gather_rows_f32(ids.data_ptr(), table.data_ptr(), out.data_ptr(), ids.numel(), 300, V, st))
torch.cuda.synchronize()
print(point, 'tokens', ids.numel(), 'table MB', V * 1200 / 1e6, 'algorithmic read bytes per launch', ids.numel() * 1208)
