"""Host-side checks that need no GPU: the C-ABI library loads and exports every declared symbol, the drop-in
package mirrors the reference's module API / state_dict, the product fails LOUDLY without a GPU (no CPU fallback),
and the launcher really runs the reference's unchanged train.py against the engine's model package."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/src'


def test_library_exports_every_declared_symbol():
    from news_recommendation_amd import _capi
    hdr = open(os.path.join(ROOT, 'include', 'nr_engine.h')).read()
    declared = set(re.findall(r'^\s*(?:int|int64_t|const char\*)\s+(nr_\w+)\s*\(', hdr, flags=re.M))
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    lib = _capi.load()                                   # built by __graft_entry__.build()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nr_version() >= 1 and lib.nr_supported_seq_len(20) == 1 and lib.nr_supported_seq_len(7) == 0
    assert lib.nr_additive_bwd_grid(17, 20) == 2 and lib.nr_additive_bwd_grid(10, 50) == 10 and lib.nr_additive_bwd_grid(4097, 50) == 1025      # k_pool2.h: 16 titles per workgroup; 4 fifty-token sequences from 2048 up
    # argument validation happens before any device work, so it can be exercised without a GPU
    assert lib.nr_gather_rows_f32(None, None, None, 5, 300, 10, None) == -2
    assert b'nr_gather_rows_f32' in lib.nr_last_error()


def test_dropin_state_dict_matches_reference_keys():
    from news_recommendation_amd.dropin.model.NRMS import NRMS
    from oracle.make_golden import make_cfg
    from oracle.nrms_numpy import nrms_param_shapes
    m = NRMS(make_cfg(777, 300, 15, 200, 50, 20, 0.2))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == nrms_param_shapes(777)
    assert [n for n, _ in m.named_children()] == ['news_encoder', 'user_encoder', 'click_predictor']
    pre = torch.randn(777, 300)
    m2 = NRMS(make_cfg(777, 300, 15, 200, 50, 20, 0.2), pre)
    assert torch.equal(m2.news_encoder.word_embedding.weight.detach(), pre)      # row 0 NOT zeroed
    assert m2.news_encoder.word_embedding.padding_idx == 0


def test_naml_lstur_dropin_state_dicts_match_reference_keys():
    """Key names and shapes of the drop-in NAML / LSTUR equal the seeded parameter sets that oracle/make_golden_naml_lstur.py loaded
    into the REFERENCE's own modules with load_state_dict (strict), i.e. the reference's state_dict layout."""
    from news_recommendation_amd.dropin.model.NAML import NAML
    from news_recommendation_amd.dropin.model.LSTUR import LSTUR
    from oracle.naml_torch import random_naml_params
    from oracle.lstur_torch import random_lstur_params

    class NamlCfg:
        dataset_attributes = {"news": ['category', 'subcategory', 'title', 'abstract'], "record": []}
        num_words, word_embedding_dim, num_categories, category_embedding_dim = 321, 300, 275, 100
        num_filters, window_size, query_vector_dim, dropout_probability = 300, 3, 200, 0.2
    m = NAML(NamlCfg)
    want = random_naml_params(0, 321, 300, 275, 100, 300, 3, 200)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in want.items()}
    assert [n for n, _ in m.named_children()] == ['news_encoder', 'user_encoder', 'click_predictor']
    te = m.news_encoder.text_encoders
    assert te['title'].word_embedding is te['abstract'].word_embedding                       # one shared table, two keys

    for method, width in (('ini', 900), ('con', 450)):
        class LsturCfg:
            dataset_attributes = {"news": ['category', 'subcategory', 'title'], "record": ['user', 'clicked_news_length']}
            num_words, word_embedding_dim, num_categories, num_users = 321, 300, 275, 77
            num_filters, window_size, query_vector_dim, dropout_probability, masking_probability = 300, 3, 200, 0.2, 0.5
            long_short_term_method = method
        m = LSTUR(LsturCfg)
        want = random_lstur_params(0, 321, 300, 275, 77, 300, 3, 200, method)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in want.items()}
        assert tuple(m.user_embedding.weight.shape) == (77, width)
    # geometry knobs away from the tuned instantiation construct (general-geometry path, tests/test_generic_gpu.py); what neither path runs raises
    class Other(NamlCfg):
        num_filters, window_size, word_embedding_dim, query_vector_dim = 128, 5, 100, 256
    assert tuple(NAML(Other).news_encoder.text_encoders['title'].CNN.weight.shape) == (128, 1, 5, 100)
    with pytest.raises(NotImplementedError):
        class Bad(NamlCfg):
            window_size = 4                  # the reference asserts an odd window (LSTUR/news_encoder.py:23)
        NAML(Bad)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    from news_recommendation_amd.dropin.model.NRMS import NRMS
    from oracle.make_golden import make_cfg
    m = NRMS(make_cfg(100, 300, 15, 200, 50, 20, 0.2))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.get_news_vector({'title': torch.zeros(2, 20, dtype=torch.long)})
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.get_user_vector(torch.zeros(2, 50, 300))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.get_prediction(torch.zeros(3, 300), torch.zeros(300))


def test_product_does_not_import_oracle_or_emulator():
    pkg = os.path.join(ROOT, 'news_recommendation_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.sh')):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
                assert 'tests/emu' not in src.replace('tests/emu/ for the CPU emulation build', ''), f
                assert 'libnr_engine_emu' not in src, f


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
@pytest.mark.skipif(torch.cuda.is_available(), reason="container-only check")
@pytest.mark.parametrize('model_name', ['NRMS', 'NAML', 'LSTUR'])
def test_launcher_runs_unchanged_reference_train_against_engine(tmp_path, model_name):
    """train.py (unchanged, read-only) is executed by the launcher; it builds the ENGINE's model, loads the data with
    the reference's own dataset.py and reaches the first forward, where the engine refuses to run without a GPU."""
    from news_recommendation_amd import synth
    synth.write_reference_dataset(str(tmp_path))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH=ROOT)
    env.pop('MODEL_NAME', None)
    p = subprocess.run([sys.executable, '-m', 'news_recommendation_amd.launcher', 'train', '--reference', REF, '--workdir', str(tmp_path),
                        '--model', model_name], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert 'Load training dataset with size 256' in out, out[-2000:]             # reference train.py:116
    assert f'news_recommendation_amd/dropin/model/{model_name}' in out, out[-2000:]       # traceback goes through OUR model
    assert 'no CPU fallback' in out, out[-2000:]


def test_grad_target_and_profile_switches():
    """Host-side switches of the table-gradient and whole-recurrence paths (no GPU needed)."""
    import torch
    from news_recommendation_amd import ops, dist as nrdist
    p = torch.nn.Parameter(torch.zeros(5, 3))
    dst, ret = ops.grad_target(p)                       # plain autograd: a fresh zeroed tensor is returned to autograd
    assert ret is dst and dst.shape == p.shape and not dst.any()
    q = torch.nn.Parameter(torch.zeros(5, 3))
    fgb = nrdist.FlatGradBuffer([p, q])
    dst, ret = ops.grad_target(p)                       # FlatGradBuffer: accumulate in place, nothing returned
    assert ret is None and dst.data_ptr() == fgb.flat.data_ptr()
    p.grad = None                                       # optimizer.zero_grad(set_to_none=True) drops the view: back to the plain path
    dst, ret = ops.grad_target(p)
    assert ret is dst
    assert not ops.profiling('nr_gru_fwd_step')
    with ops.profile():
        assert ops.profiling('nr_gru_fwd_step')
    with ops.profile(only={'nr_gru_bwd_seq'}):
        assert not ops.profiling('nr_gru_bwd_step') and ops.profiling('nr_gru_bwd_seq')
    t = ops.to_device_async(torch.arange(3), 'cpu')
    assert t.tolist() == [0, 1, 2]


def test_split_rows_and_inplace_grads():
    """ops.split_rows == slicing (values and gradients, also when one part is unused); ops.inplace_grads hands out the trainer's gradient
    views only when EVERY parameter has one."""
    import torch
    from news_recommendation_amd import ops, dist as nrdist
    x = torch.randn(7, 3, requires_grad=True)
    a, b = ops.split_rows(x, 3)
    assert torch.equal(a, x[:3]) and torch.equal(b, x[3:])
    (a.sum() * 2 + (b * b).sum()).backward()
    y = x.detach().clone().requires_grad_(True)
    (y[:3].sum() * 2 + (y[3:] * y[3:]).sum()).backward()
    assert torch.equal(x.grad, y.grad)
    x.grad = None
    a, b = ops.split_rows(x, 3)
    a.sum().backward()                                  # the unused part contributes zeros
    assert torch.equal(x.grad[:3], torch.ones(3, 3)) and not x.grad[3:].any()
    p, q, r = (torch.nn.Parameter(torch.zeros(4, 2)) for _ in range(3))
    assert ops.inplace_grads((p, q)) is None            # plain autograd
    fgb = nrdist.FlatGradBuffer([p, q])
    g = ops.inplace_grads((p, q))
    assert g is not None and g[0].data_ptr() == fgb.flat.data_ptr() and g[1].shape == q.shape
    assert ops.inplace_grads((p, q, r)) is None         # r is not the trainer's
    q.grad = None
    assert ops.inplace_grads((p, q)) is None            # zero_grad(set_to_none=True) dropped a view


def _shard_worker(rank, world, root, q):
    """One launcher rank up to (not including) the reference's train.py: per-rank working directory + checkpoint gate."""
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from news_recommendation_amd import launcher
    d = launcher.shard_workdir(root, rank, world)
    launcher.rank0_only_checkpoints(rank)
    os.chdir(d)
    with open('data/train/behaviors_parsed.tsv') as f:
        lines = f.read().splitlines()
    os.makedirs('checkpoint/NRMS', exist_ok=True)
    torch.save({'rank': rank}, f'./checkpoint/NRMS/ckpt-{10 + rank}.pth')          # what train.py:264-277 does on every rank
    q.put((rank, d, lines, sorted(os.listdir('data/train')), os.path.islink('data/val'), os.path.realpath('checkpoint')))


def test_launcher_shards_training_rows_per_rank_and_gates_checkpoints(tmp_path):
    """SURVEY 8 e2 for the unchanged train.py: two ranks, each its own working directory with HALF of behaviors_parsed.tsv (equal counts,
    disjoint, header kept), the rest of ./data shared by symlinks, one shared ./checkpoint that only rank 0 writes."""
    import multiprocessing as mp
    from news_recommendation_amd import synth
    root = str(tmp_path)
    synth.write_reference_dataset(root, n_news=60, n_train=101, n_val_impr=8, num_words=200)
    with open(os.path.join(root, 'data', 'train', 'behaviors_parsed.tsv')) as f:
        full = f.read().splitlines()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (d0, l0, files0, val0, ck0), (d1, l1, files1, val1, ck1) = res[0], res[1]
    assert d0 != d1 and l0[0] == l1[0] == full[0]
    n = (len(full) - 1) // 2
    assert len(l0) - 1 == len(l1) - 1 == n == 50
    assert l0[1:] == full[1:][0:2 * n:2] and l1[1:] == full[1:][1:2 * n:2]
    assert not set(l0[1:]) & set(l1[1:]) or len(set(full[1:])) < len(full) - 1
    assert files0 == files1 and 'news_parsed.tsv' in files0 and val0 and val1
    assert ck0 == ck1 == os.path.realpath(os.path.join(root, 'checkpoint'))
    assert os.listdir(os.path.join(root, 'checkpoint', 'NRMS')) == ['ckpt-10.pth']          # rank 1's torch.save was gated


def test_pool_flat_dispatch_by_shape(monkeypatch):
    """Which pooling levels take the flat backward kernel (ops.pool_flat_ok): sequences long enough that 48 consecutive tokens span at most
    8 of them (4 with the activation gradient), launches worth a persistent kernel; NR_POOL_FLAT=0 switches it off everywhere."""
    from news_recommendation_amd import ops
    monkeypatch.setattr(ops, '_POOL_FLAT', True)
    monkeypatch.setattr(ops, '_POOL_FLAT_MIN_TOK', 98304)
    q = dict(qdim=200)
    assert ops.pool_flat_ok(20, False, 27136, **q) and ops.pool_flat_ok(50, True, 28160, **q) and ops.pool_flat_ok(20, True, 28160, **q)
    monkeypatch.setattr(ops, '_POOL_FLAT_SHORT', True)
    assert ops.pool_flat_ok(4, False, 28160, **q) and not ops.pool_flat_ok(3, False, 40000, **q)     # the four views of a NAML news item: 13 sequences in 48 tokens (second slot tile, round 6)
    monkeypatch.setattr(ops, '_POOL_FLAT_SHORT', False)
    assert not ops.pool_flat_ok(4, False, 28160, **q)            # NR_POOL_FLAT_VIEWS=0: the sequence-shaped kernel at the view level
    assert ops.pool_flat_ok(7, False, 20000, **q) and not ops.pool_flat_ok(15, True, 20000, **q) and ops.pool_flat_ok(16, True, 20000, **q)
    assert not ops.pool_flat_ok(50, False, 512, **q)             # 512 click histories: the sequence-shaped kernel is faster
    assert ops.pool_flat_ok(50, False, **q)                      # shape-only question (operands packed before the batch is known)
    # the flat kernel keeps 200 rows of the projection matrix in LDS: query_vector_dim 201 .. 208 stays on the sequence-shaped kernels
    assert ops.pool_flat_ok(20, False, 27136, qdim=1) and not ops.pool_flat_ok(20, False, 27136, qdim=201) and not ops.pool_flat_ok(50, True, 28160, qdim=208)
    with pytest.raises(TypeError):
        ops.pool_flat_ok(20, False, 27136)                       # qdim is keyword-only and required: no call site can forget it
    monkeypatch.setattr(ops, '_POOL_FLAT', False)
    assert not ops.pool_flat_ok(20, False, 27136, **q)
