"""Training driver with the reference's ``train()`` semantics (src/train.py:67-279) on GPU-resident data.

    python -m news_recommendation_amd.train_fast --workdir RUN_DIR --model NRMS [--reference /path/to/reference/src] [--set k=v ...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m news_recommendation_amd.train_fast ...

Same files (``./data/train/{behaviors_parsed,news_parsed}.tsv``, ``pretrained_word_embedding.npy``, ``./data/val``), same
hyper-parameters (the reference's ``config.py`` with ``--reference``, else default_config.py), same loop: Adam(lr), CrossEntropy
against class 0, ``num_epochs * len(dataset) // batch_size`` iterations with reshuffling, loss lines every
``num_batches_show_loss``, validation every ``num_batches_validate`` on ./data/val (max_count 200000) with early stopping
(patience 5) on -AUC, and checkpoints ``./checkpoint/<MODEL>/ckpt-<step>.pth`` in the reference's format (resumable by either
trainer, loadable by ``evaluate.py``).  Different on purpose: the input pipeline (data_fast.py), validation (evaluate_fast.py),
and data parallelism (each rank a shard of the samples, one flat RCCL all-reduce per step; rank 0 validates and saves).
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch


class EarlyStopping:
    """src/train.py:27-51."""

    def __init__(self, patience=5):
        self.patience, self.counter, self.best_loss = patience, 0, float('inf')

    def __call__(self, val_loss):
        if val_loss < self.best_loss:
            self.counter, self.best_loss = 0, val_loss
            return False, True
        self.counter += 1
        return self.counter >= self.patience, False


def checkpoint_safe_globals():
    """The numpy types a checkpoint of the reference's format needs beside tensors: its 'early_stop_value' is a numpy scalar
    (train.py:268-275), which torch >= 2.6's default weights_only=True unpickler rejects unless these are allow-listed."""
    try:
        from numpy._core import multiarray as _ma
    except ImportError:                      # numpy < 2
        from numpy.core import multiarray as _ma
    return [_ma.scalar, np.dtype] + [type(np.dtype(t)) for t in (np.float64, np.float32, np.float16, np.int64, np.int32, np.bool_)]


def load_checkpoint(path, device):
    """A checkpoint of this run directory, written by this trainer or by the reference's train.py (the launcher path): weights_only stays
    on, with checkpoint_safe_globals() allowed."""
    with torch.serialization.safe_globals(checkpoint_safe_globals()):
        return torch.load(path, map_location=device, weights_only=True)


def latest_checkpoint(directory):
    """src/train.py:54-64."""
    if not os.path.exists(directory):
        return None
    ck = {int(x.split('.')[-2].split('-')[-1]): x for x in os.listdir(directory)}
    return os.path.join(directory, ck[max(ck)]) if ck else None


def load_config(model_name, reference_src, overrides):
    if reference_src:
        os.environ['MODEL_NAME'] = model_name
        sys.path.insert(0, os.path.abspath(reference_src))
        sys.dont_write_bytecode = True
        base = getattr(importlib.import_module('config'), f'{model_name}Config')
    else:
        from news_recommendation_amd import default_config
        base = getattr(default_config, f'{model_name}Config')
    cfg = type(f'{model_name}Config', (base,), {})
    for kv in overrides:
        k, v = kv.split('=', 1)
        old = getattr(base, k)
        setattr(cfg, k, type(old)(v) if not isinstance(old, bool) else v == 'True')
    return cfg


def train(model_name, config, workdir='.', max_steps=None, log=print):
    from news_recommendation_amd import dist as nrdist, evaluate_fast
    from news_recommendation_amd.optim import EngineAdam
    from news_recommendation_amd.data_fast import TrainData, forward_batch
    import torch.distributed as dist
    rank, world, local = nrdist.init_from_env()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    os.chdir(workdir)
    Model = getattr(importlib.import_module(f'news_recommendation_amd.dropin.model.{model_name}'), model_name)
    try:
        pre = torch.from_numpy(np.load('./data/train/pretrained_word_embedding.npy')).float()
    except FileNotFoundError:
        pre = None
    model = Model(config, pre).to(device)
    nrdist.broadcast_parameters(model)
    data = TrainData('data/train/behaviors_parsed.tsv', 'data/train/news_parsed.tsv', config, device, rank, world)
    n_total = data.n_total
    if rank == 0:
        log(f"Load training dataset with size {n_total}.")
    # Adam with the reference's hyper-parameters (train.py:127-128) on the engine's flat buffers: fused update kernel, gradient exchange
    # overlapped with the backward, LSTUR's user table as a row-sparse table; state_dict() keeps torch.optim.Adam's format
    optimizer = EngineAdam(model, lr=config.learning_rate, row_sparse=('user_embedding.weight',) if model_name == 'LSTUR' else ())
    early_stopping = EarlyStopping()
    step = 0
    ckdir = os.path.join('./checkpoint', model_name)
    if rank == 0:
        os.makedirs(ckdir, exist_ok=True)
    path = latest_checkpoint(ckdir)
    if path is not None:
        log(f"Load saved parameters in {path}")
        ck = load_checkpoint(path, device)
        early_stopping(ck['early_stop_value'])
        step = ck['step']
        model.load_state_dict(ck['model_state_dict'])
        optimizer.load_state_dict(ck['optimizer_state_dict'])
    model.train()
    per_rank_batch = config.batch_size                       # weak scaling: config.batch_size impressions per GPU and step
    # the iteration count derives from GLOBAL quantities only: shard sizes differ by up to one sample, and a rank with an extra iteration
    # would wait forever in the gradient exchange (or in the validation broadcast)
    n_iter = config.num_epochs * (n_total // world) // per_rank_batch
    if max_steps is not None:
        n_iter = min(n_iter, max_steps)
    it = data.batches(per_rank_batch)
    # train.py:225,241-244 appends loss.item() every step (a device->host sync per step); here the running sum and the last 256 losses
    # stay on the device and are read only on the steps that print
    loss_sum = torch.zeros((), dtype=torch.float64, device=device)
    recent = torch.zeros(256, dtype=torch.float32, device=device)
    t0, seen = time.time(), 0
    val_plan = None
    # LSTUR's persistent GRU sweeps can fail (a bounded XCD-local wait gives up: csrc/k_xcd.h).  The device never trains on such a sweep -- sticky
    # fault words gate every optimiser kernel from the failed step on (csrc/k_optim.h) -- and the host, at its next look (the steps that print or
    # validate synchronise anyway), repeats the skipped steps on the step-per-launch kernels: `window` keeps their batches until then.
    guard = model_name == 'LSTUR'
    if guard:
        optimizer.attach_fault_words()
    window = []                           # (iteration, optimiser step index, batch) since the last look

    def train_step(b):
        loss = forward_batch(model, b, loss=True)           # scorer + CrossEntropyLoss against class 0 (train.py:205-206) as one kernel pair
        loss.backward()                   # table all-reduce starts inside; gradients accumulate into the optimiser's flat buffer
        optimizer.step()                  # remaining exchange + fused Adam (clears the gradients: no zero_grad pass)
        return loss.detach()

    for i in range(1, n_iter + 1):
        try:
            b = next(it)
        except StopIteration:
            it = data.batches(per_rank_batch)
            b = next(it)
        step += 1
        t_step = optimizer.t + 1
        ld = train_step(b)
        recent[(i - 1) % 256] = ld
        if guard:
            window.append([i, t_step, b, ld])
        else:
            loss_sum += ld
        seen += per_rank_batch * world
        look = i % config.num_batches_show_loss == 0 or i == n_iter or i % config.num_batches_validate == 0
        if guard and look:
            first = optimizer.rewind_after_fault()          # synchronises; every rank gets the same answer
            if first is not None:
                os.environ['NR_GRU_PERSIST'] = '0'          # read per call by the library: from here on one launch per GRU step
                redo = [e for e in window if e[1] >= first]
                if rank == 0:
                    log(f"persistent GRU sweep failed at optimiser step {first}: that update and the {max(len(redo) - 1, 0)} after it were skipped "
                        f"on the device; repeating {len(redo)} step(s) with NR_GRU_PERSIST=0")
                for e in redo:
                    e[3] = ld = train_step(e[2])
                    recent[(e[0] - 1) % 256] = ld
            if window:
                loss_sum += torch.stack([e[3] for e in window]).double().sum()
            window.clear()
        if i % config.num_batches_show_loss == 0 or i == n_iter:
            if rank == 0:        # same three numbers as train.py:241-244: current, mean over all steps, mean over the latest 256
                log(f"Time {time.strftime('%H:%M:%S', time.gmtime(time.time() - t0))}, batches {i}, current loss {float(ld):.4f}, "
                    f"average loss: {float(loss_sum) / i:.4f}, latest average loss: {float(recent[:min(i, 256)].mean()):.4f}, "
                    f"{seen / (time.time() - t0):.0f} impressions/s")
        if i % config.num_batches_validate == 0:
            stop = torch.zeros(2, device=device)           # [stop, save]: rank 0 decides, every rank learns both
            if rank == 0:
                model.eval()
                if val_plan is None:      # the validation files do not change during a run: parse them once
                    val_plan = evaluate_fast.build_plan('./data/val', config.dataset_attributes['news'], config.num_clicked_news_a_user,
                                                        max_count=200000)
                auc, mrr, n5, n10 = evaluate_fast.evaluate(model, './data/val', config.num_workers, 200000, plan=val_plan)
                model.train()
                log(f"Time {time.strftime('%H:%M:%S', time.gmtime(time.time() - t0))}, batches {i}, validation AUC: {auc:.4f}, "
                    f"validation MRR: {mrr:.4f}, validation nDCG@5: {n5:.4f}, validation nDCG@10: {n10:.4f}, ")
                early_stop, get_better = early_stopping(-auc)
                if early_stop:
                    log('Early stop.')
                    stop[0] += 1
                elif get_better:
                    stop[1] += 1
            if world > 1:
                dist.broadcast(stop, src=0)
            flags = stop.tolist()
            if flags[1] > 0:
                # sharded Adam moments (NR_TABLE_RS): collecting them is a collective, so EVERY rank takes part before rank 0 writes the file
                optimizer.gather_state()
                if rank == 0:
                    torch.save({'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(), 'step': step,
                                'early_stop_value': -auc}, f"./checkpoint/{model_name}/ckpt-{step}.pth")
            if flags[0] > 0:
                break
    torch.cuda.synchronize()
    if guard:
        optimizer.detach_fault_words()
    return {'steps': i, 'impressions_per_s': seen / max(time.time() - t0, 1e-9), 'last_loss': float(ld.item()), 'model': model, 'optimizer_steps': optimizer.t}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--workdir', default='.')
    ap.add_argument('--model', default=os.environ.get('MODEL_NAME', 'NRMS'), choices=['NRMS', 'NAML', 'LSTUR'])
    ap.add_argument('--reference', default=None, help="reference checkout's src/ directory: use its config.py")
    ap.add_argument('--set', nargs='*', default=[], metavar='KNOB=VALUE', help='override config attributes')
    ap.add_argument('--max-steps', type=int, default=None)
    a = ap.parse_args(argv)
    train(a.model, load_config(a.model, a.reference, a.set), a.workdir, a.max_steps)


if __name__ == '__main__':
    main()
