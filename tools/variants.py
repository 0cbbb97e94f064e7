"""A/B the mhsa forward variants on one box, interleaved rounds.  The library reads its knobs once per process, so every measurement is
its own `tools/prof_kernel.py` process.  Usage: python tools/variants.py [KNOB=a,b ...]   (default: NR_MHSA_VARIANT=2,81)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
knobs = [a.split('=', 1) for a in sys.argv[1:]] or [['NR_MHSA_VARIANT', '2,81']]
kernels = ('mhsa_infer', 'mhsa_train')
for rnd in range(3):
    for name, values in knobs:
        for v in values.split(','):
            env = dict(os.environ, **{name: v})
            out = []
            for k in kernels:
                r = subprocess.run([sys.executable, os.path.join(HERE, 'prof_kernel.py'), k], env=env, capture_output=True, text=True)
                out.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else f'{k} FAILED')
            print(rnd, f'{name}={v}', ' | '.join(out), flush=True)
