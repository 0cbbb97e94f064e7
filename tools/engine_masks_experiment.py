#!/usr/bin/env python
"""Does the engine's counter-based dropout RNG explain the AUC difference of the statistical training-parity leg?  (ANALYSIS, CPU only.)

The oracle (fp32 torch restatement of the reference, torch.optim.Adam) is trained on the NRMS fixture task with the masks the ENGINE would draw:
the generator of csrc/nr_common.h (drop_words / drop_keep4) restated in numpy (checked against nr_dropout_mask of the emulator build), the
seeds drawn exactly like bench.train_parity_fixture's engine runs (torch.manual_seed(1000 + s); one 62-bit draw per step), the element
numbering of the stacked title batch (candidates b * C + c, then clicked B * C + b * N + n).  If these runs land where the reference's
F.dropout runs land, the masks are not the cause.

    python tools/engine_masks_experiment.py [--seeds 8] [--procs 3]"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
U = np.uint32


def mix32(x):
    x = x ^ (x >> U(16)); x = x * U(0x7feb352d); x = x ^ (x >> U(15)); x = x * U(0x846ca68b); x = x ^ (x >> U(16))
    return x


def keep_mask(n_elem, p, seed, site):
    """keep flags (float32 0 / 1) of elements 0 .. n_elem - 1 (n_elem % 4 == 0): csrc/nr_common.h drop_words + drop_keep4."""
    with np.errstate(over='ignore'):
        k0, k1 = U(seed & 0xFFFFFFFF), U((seed >> 32) & 0xFFFFFFFF)
        t = p * 65536.0 + 0.5
        thresh = U(65535 if t >= 65535.0 else int(t))
        quad = np.arange(n_elem // 4, dtype=np.uint64)
        lo, hi = (quad & np.uint64(0xFFFFFFFF)).astype(U), (quad >> np.uint64(32)).astype(U)
        r0 = mix32(lo ^ ((hi << U(16)) | (hi >> U(16))) ^ k0 ^ U((site * 0x85EBCA77) & 0xFFFFFFFF))
        x = (r0 ^ k1) * U(0x9E3779B1)
        r1 = x ^ (x >> U(15))
        t16 = thresh << U(16)
        m = np.stack([(r0 << U(16)) >= t16, r0 >= t16, (r1 << U(16)) >= t16, r1 >= t16], axis=1)
    return m.reshape(-1).astype(np.float32)


def run(args):
    seed, threads = args
    import torch
    from oracle import train_parity as tp
    from oracle.nrms_torch import OracleNRMS
    torch.set_num_threads(threads)
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_parity', 'nrms.npz'))
    task = tp.task_from_arrays(z)
    st0 = tp.init_state(task["num_words"])
    m = OracleNRMS(task["num_words"], 300, 15, 200, 0.2)
    m.load_state_dict(st0)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss()
    B = task["B"]
    C, N, L = task["cand_ids"].shape[2], task["click_ids"].shape[2], task["cand_ids"].shape[3]
    T = B * (C + N)
    y = torch.zeros(B, dtype=torch.long)
    torch.manual_seed(1000 + seed)
    losses = []
    for i in range(task["steps"]):
        dseed = int(torch.randint(0, 2 ** 62, (1,)).item())          # ops.new_seed()
        m1 = keep_mask(T * L * 300, 0.2, dseed, 1).reshape(T, L, 300)
        m2 = keep_mask(T * L * 300, 0.2, dseed, 2).reshape(T, L, 300)
        keeps = []
        for j in range(C + N):
            idx = np.arange(B) * C + j if j < C else B * C + np.arange(B) * N + (j - C)
            keeps.append({'title1': torch.from_numpy(m1[idx]), 'title2': torch.from_numpy(m2[idx])})
        loss = crit(m(tp.as_lists(task["cand_ids"][i]), tp.as_lists(task["click_ids"][i]), keeps), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    met = tp.eval_metrics(task, tp.oracle_eval_scores(task, {k: v.detach() for k, v in m.state_dict().items()}))
    return seed, [float(x) for x in met], float(np.mean(losses[-10:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=8)
    ap.add_argument('--procs', type=int, default=3)
    ap.add_argument('--threads', type=int, default=2)
    a = ap.parse_args()
    # the numpy restatement against the engine's own export (emulator build of the same sources)
    from tests.backends import EmuBackend
    from tests.kernel_checks import export_mask
    be = EmuBackend()
    for seed, site in ((0x0123456789abcdef, 1), (0x2fedcba987654321, 2), (77, 3)):
        assert np.array_equal(export_mask(be, 40000, 0.2, seed, site), keep_mask(40000, 0.2, seed, site)), (seed, site)
    print("numpy generator == nr_dropout_mask (emulator build)", flush=True)
    with mp.get_context('spawn').Pool(a.procs) as pool:
        res = {s: (met, l10) for s, met, l10 in pool.imap_unordered(run, [(s, a.threads) for s in range(a.seeds)])}
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_parity', 'nrms.npz'))
    auc = np.array([res[s][0][0] for s in sorted(res)])
    ref = z['ref_metrics'][:, 0]
    se = math.sqrt(auc.var(ddof=1) / len(auc) + ref.var(ddof=1) / len(ref))
    print(json.dumps({"oracle_with_engine_masks_auc": [round(float(x), 4) for x in auc], "last10_loss": [round(res[s][1], 4) for s in sorted(res)],
                      "mean": float(auc.mean()), "reference_mean": float(ref.mean()), "diff": float(auc.mean() - ref.mean()), "stderr": se,
                      "z": float((auc.mean() - ref.mean()) / se)}))


if __name__ == '__main__':
    main()
