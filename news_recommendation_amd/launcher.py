"""Run the reference's UNCHANGED entry scripts (src/train.py, src/evaluate.py) on the MI355X engine.

    python -m news_recommendation_amd.launcher train    --reference /path/to/reference/src --workdir RUN_DIR
    python -m news_recommendation_amd.launcher evaluate --reference /path/to/reference/src --workdir RUN_DIR

How the drop-in works:
  * sys.path = [<this package>/dropin, <repo root>, <reference>/src, ...]: ``import model.NRMS`` resolves to the
    engine's ``dropin/model`` package (same class names / signatures / state_dict keys), while ``config``,
    ``dataset``, ``evaluate`` and ``train`` resolve to the reference's read-only files;
  * three shims for this container's package set: a no-op ``torch.utils.tensorboard.SummaryWriter``
    when tensorboard is not installed, ``numpy.Inf`` (removed in NumPy 2, used at train.py:31), and ``torch.load``
    defaulting to ``weights_only=False`` as it did when the reference was written (its checkpoints hold a numpy scalar);
  * cwd = RUN_DIR, which holds ./data (reference formats), ./checkpoint, ./runs -- every reference path is
    cwd-relative;
  * under torchrun (WORLD_SIZE > 1) each rank pins its own GPU (so the reference's hard-coded ``cuda:0`` is the local
    device), replicas are synchronised and gradients are all-reduced over RCCL by post-accumulate hooks -- train.py
    needs no DDP wrapper.
"""
import argparse
import os
import runpy
import sys
import types


def install_shims():
    import numpy
    if not hasattr(numpy, 'Inf'):
        numpy.Inf = numpy.inf
    # torch >= 2.6 defaults torch.load(weights_only=True); the reference's checkpoints (train.py:268-275) carry a numpy scalar
    # ('early_stop_value') and are loaded with a bare torch.load(path) (train.py:149, evaluate.py:287), which that default rejects.
    # weights_only stays on for every load of the process: only the numpy scalar / dtype types that value needs are allow-listed.
    import torch
    from news_recommendation_amd.train_fast import checkpoint_safe_globals
    torch.serialization.add_safe_globals(checkpoint_safe_globals())
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        mod = types.ModuleType('torch.utils.tensorboard')

        class SummaryWriter:               # the reference only calls add_scalar / add_scalars (train.py:238-255)
            def __init__(self, *a, **k): pass
            def add_scalar(self, *a, **k): pass
            def add_scalars(self, *a, **k): pass
            def close(self): pass
        mod.SummaryWriter = SummaryWriter
        sys.modules['torch.utils.tensorboard'] = mod


def _patch_dp(model_name):
    """Data parallelism without touching train.py: wrap the engine model's .to() so every new model is
    broadcast from rank 0 and gets the gradient all-reduce hooks."""
    import importlib
    from news_recommendation_amd import dist as nrdist
    rank, world, local = nrdist.init_from_env()
    if world <= 1:
        return
    cls = getattr(importlib.import_module(f'model.{model_name}'), model_name)
    orig_to = cls.to

    def to(self, *a, **k):
        out = orig_to(self, *a, **k)
        if not getattr(out, '_nr_dp', False) and next(out.parameters()).is_cuda:
            nrdist.broadcast_parameters(out)
            nrdist.attach_grad_allreduce(out)
            out._nr_dp = True
        return out
    cls.to = to


def _fast_evaluate_main(model_name):
    """`evaluate.py`'s __main__ block (src/evaluate.py:275-294) with the batched driver: newest checkpoint -> test metrics."""
    import importlib
    import torch
    import evaluate as ref_eval                       # the reference's module: config, latest_checkpoint live there / in train.py
    from news_recommendation_amd import evaluate_fast
    config = ref_eval.config
    Model = getattr(importlib.import_module(f'model.{model_name}'), model_name)
    model = Model(config).to(ref_eval.device)
    ckpt_dir = os.path.join('./checkpoint', model_name)
    ckpts = {int(f.split('.')[-2].split('-')[-1]): f for f in os.listdir(ckpt_dir)} if os.path.isdir(ckpt_dir) else {}
    if not ckpts:
        print('No checkpoint file found!')
        return
    path = os.path.join(ckpt_dir, ckpts[max(ckpts)])
    print(f"Load saved parameters in {path}")
    model.load_state_dict(torch.load(path)['model_state_dict'])
    model.eval()
    auc, mrr, ndcg5, ndcg10 = evaluate_fast.evaluate(model, './data/test', config.num_workers)
    print(f'AUC: {auc:.4f}\nMRR: {mrr:.4f}\nnDCG@5: {ndcg5:.4f}\nnDCG@10: {ndcg10:.4f}')


def narrow_visibility(env):
    """One GPU per rank BEFORE torch is imported: the reference hard-codes ``cuda:0`` in 13 module globals, so the only
    way to run its unchanged files on the rank's own GPU is to make that GPU the process's device 0.  If the caller already restricts
    visibility to a list (``HIP_VISIBLE_DEVICES=4,5,6,7``) the rank takes the LOCAL_RANK-th entry of that list; a single pre-set
    entry per process is left alone.  ``dist.local_device_index()`` then resolves the local device index to 0."""
    if env.get('WORLD_SIZE', '1') == '1' or 'LOCAL_RANK' not in env:
        return
    lr = int(env['LOCAL_RANK'])
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        cur = env.get(var)
        if cur:
            devs = [d for d in cur.split(',') if d != '']
            if len(devs) > 1:
                if lr >= len(devs):
                    raise RuntimeError(f"LOCAL_RANK={lr} but {var}={cur!r} lists only {len(devs)} devices")
                env[var] = devs[lr]
            return
    env['HIP_VISIBLE_DEVICES'] = str(lr)


def shard_workdir(workdir, rank, world):
    """Data parallelism for the UNCHANGED train.py: the reference builds its DataLoader from ./data/train/behaviors_parsed.tsv
    with shuffle=True and no DistributedSampler (train.py:113-124), so every rank gets its OWN working directory whose behaviors_parsed.tsv
    holds that rank's 1/world of the rows (row i goes to rank i % world; every shard is cut to floor(n / world) rows, because the step count
    derives from len(dataset) (train.py:166) and a rank with one more step would wait forever in the gradient exchange).  Everything else
    of ./data is shared through symlinks; ./checkpoint is rank 0's (all ranks resume from the same file); returns the directory."""
    workdir = os.path.abspath(workdir)
    if world <= 1:
        return workdir
    dst = os.path.join(workdir, f'.dp_rank{rank}_of_{world}')
    src_train = os.path.join(workdir, 'data', 'train')
    os.makedirs(os.path.join(dst, 'data', 'train'), exist_ok=True)

    def link(src, name):
        if not os.path.lexists(name):
            os.symlink(src, name)
    for entry in os.listdir(os.path.join(workdir, 'data')):
        if entry != 'train':
            link(os.path.join(workdir, 'data', entry), os.path.join(dst, 'data', entry))
    for entry in os.listdir(src_train):
        if entry != 'behaviors_parsed.tsv':
            link(os.path.join(src_train, entry), os.path.join(dst, 'data', 'train', entry))
    for d in ('checkpoint', 'runs'):
        os.makedirs(os.path.join(workdir, d), exist_ok=True)
        link(os.path.join(workdir, d), os.path.join(dst, d))
    with open(os.path.join(src_train, 'behaviors_parsed.tsv')) as f:
        header = f.readline()
        rows = f.readlines()
    per_rank = len(rows) // world
    tmp = os.path.join(dst, 'data', 'train', f'.behaviors_parsed.tsv.{os.getpid()}')
    with open(tmp, 'w') as f:
        f.write(header)
        f.writelines(rows[rank:per_rank * world:world])
    os.replace(tmp, os.path.join(dst, 'data', 'train', 'behaviors_parsed.tsv'))
    return dst


def rank0_only_checkpoints(rank):
    """Every rank runs train.py's validation + checkpoint logic on identical weights (train.py:247-277); only rank 0's torch.save reaches
    the shared ./checkpoint directory -- N concurrent writers of one path would tear the file."""
    if rank == 0:
        return
    import torch
    if not getattr(torch.save, '_nr_rank_gate', False):
        def save(*a, **k):
            return None
        save._nr_rank_gate = True
        torch.save = save


def apply_overrides(model_name, overrides):
    """``--set knob=value``: override attributes of the reference's config class IN MEMORY (src/config.py is the knob surface; its sizes
    -- num_words, num_users, ... -- are meant to be hand-edited after preprocessing, README.md:62; the file itself stays untouched).
    The reference's scripts pick the class up from sys.modules (train.py:19, evaluate.py:16, dataset.py:11)."""
    if not overrides:
        return
    import importlib
    cfg = getattr(importlib.import_module('config'), f'{model_name}Config')
    for kv in overrides:
        k, v = kv.split('=', 1)
        old = getattr(cfg, k)
        setattr(cfg, k, (v == 'True') if isinstance(old, bool) else type(old)(v))


def run(script, reference_src, workdir, model_name='NRMS', fast_eval=False, overrides=()):
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    reference_src = os.path.abspath(reference_src)
    narrow_visibility(os.environ)
    os.environ['MODEL_NAME'] = model_name
    sys.dont_write_bytecode = True            # the reference tree is read-only
    for p in (reference_src, repo, os.path.join(here, 'dropin')):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    install_shims()
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if script == 'train' and world > 1:
        workdir = shard_workdir(workdir, rank, world)
        rank0_only_checkpoints(rank)
    os.chdir(workdir)
    apply_overrides(model_name, overrides)
    _patch_dp(model_name)
    if fast_eval:
        # the batched evaluation driver behind the reference's own name: train.py's `from evaluate import evaluate` (train.py:12)
        # then validates with it; `launcher evaluate --fast-eval` runs evaluate.py's main logic on it
        import evaluate as ref_eval
        from news_recommendation_amd import evaluate_fast
        ref_eval.evaluate = evaluate_fast.evaluate
        if script == 'evaluate':
            return _fast_evaluate_main(model_name)
    runpy.run_path(os.path.join(reference_src, script + '.py'), run_name='__main__')


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('script', choices=['train', 'evaluate'])
    ap.add_argument('--reference', default=os.environ.get('NR_REFERENCE_SRC', '/root/reference/src'))
    ap.add_argument('--workdir', default='.')
    ap.add_argument('--model', default=os.environ.get('MODEL_NAME', 'NRMS'), choices=['NRMS', 'NAML', 'LSTUR'])
    ap.add_argument('--fast-eval', action='store_true',
                    help="use the batched evaluation driver (evaluate_fast.py) instead of the reference's per-impression loop")
    ap.add_argument('--set', nargs='*', default=[], metavar='KNOB=VALUE', help="override attributes of the reference's config class in memory")
    a = ap.parse_args(argv)
    run(a.script, a.reference, a.workdir, a.model, a.fast_eval, a.set)


if __name__ == '__main__':
    main()
