"""End-to-end training throughput INCLUDING the input pipeline: reference-format files on disk -> data_fast.TrainData ->
train_fast.train.  Usage: python tools/e2e_train_bench.py [MODEL] [BATCH] [STEPS]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recommendation_amd import synth, train_fast
model = sys.argv[1] if len(sys.argv) > 1 else 'NRMS'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
d = tempfile.mkdtemp()
t0 = time.time()
synth.write_reference_dataset(d, n_news=65238, n_users=50000, n_train=B * 24, n_val_impr=100, num_words=70976)
print('dataset written in %.1f s' % (time.time() - t0), flush=True)
cfg = train_fast.load_config(model, None, [f'batch_size={B}', 'num_batches_show_loss=20', 'num_batches_validate=100000', 'num_epochs=100'])
t0 = time.time()
r = train_fast.train(model, cfg, d, max_steps=10, log=lambda s: None)          # warm-up incl. file parsing
print('parse + 10 warm-up steps: %.1f s' % (time.time() - t0), flush=True)
os.chdir('/')
r = train_fast.train(model, cfg, d, max_steps=steps, log=print)
print(f'E2E {model} B={B}: {r["impressions_per_s"]:.0f} impressions/s over {r["steps"]} steps (includes re-parsing the files: see the per-line rate above)')
