"""profiles/traffic.json from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, MI355X_MICROARCH.md 'rocprofv3 PMC slots').
Usage: python tools/pmc_traffic.py OUTDIR   (expects OUTDIR/pmc_<tag>_{fetch,write}/ with counter_collection CSVs; tags below).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: raw units are KiB; FETCH_SIZE is doubled per the guide's gfx950 calibration (128-B requests
tallied at 64 B); WRITE_SIZE is uncalibrated and taken as is.  Entries are keyed on the hash of the kernel sources and the workload so
that bench.py only reports a figure measured on the code it runs."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
out = sys.argv[1]
tags = {'proj_train': ('nr_qkv_proj_fwd[S=20]', 'qkv_proj'), 'attn_fwd': ('nr_attn_fwd[S=20]', 'attn_fwd_kernel'), 'attn_pool_fwd': ('nr_attn_pool_fwd[S=20]', 'attn_fwd_kernel'), 'attn_bwd_hm': ('nr_attn_bwd[S=20]', 'attn_bwd'),
        'additive_bwd': ('nr_additive_bwd[S=20] (sequence-shaped)', 'pool2_bwd'), 'pool_flat': ('nr_additive_bwd[S=20]', 'pool3_bwd'), 'mhsa_infer': ('nr_mhsa_fwd[S=20]', 'mhsa_fwd2'),
        # NAML-shaped stand-alone launches (tools/prof_kernel.py: 28,160 sequences of 50 rows)
        'pool_fwd_flat50': ('nr_additive_fwd[abstract]', 'pool4_fwd'), 'pool_flat50_act': ('nr_additive_bwd[abstract]', 'pool3_bwd'), 'cgemm_dgrad50': ('nr_conv3_dgrad[abstract]', 'conv_gemm'),
        'conv_abs': ('nr_conv3_fwd[abstract]', 'conv3_kernel')}
NAML_TAGS = ('pool_fwd_flat50', 'pool_flat50_act', 'cgemm_dgrad50', 'conv_abs')
SRC = sys.argv[2] if len(sys.argv) > 2 else 'profiles/r03_pmc_traffic.txt'


def avg(d, sub, counter):
    v = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r['Kernel_Name'] and r['Counter_Name'] == counter:
                v.append(float(r['Counter_Value']))
    return sum(v) / len(v) if v else None


res = {"_note": __doc__.split('\n', 1)[1].strip()}
for tag, (name, sub) in tags.items():
    f, w = avg(f'{out}/pmc_{tag}_fetch', sub, 'FETCH_SIZE'), avg(f'{out}/pmc_{tag}_write', sub, 'WRITE_SIZE')
    if f is None or w is None:
        print('missing', tag, f, w)
        continue
    res[name] = {"fetch_kib_raw": f, "write_kib_raw": w, "bytes": int((2 * f + w) * 1024), "source": SRC,
                 "source_hash": bench.kernel_source_hash(), "workload": "NAML/small/B512" if tag in NAML_TAGS else "NRMS/small/B512"}
    print(f"{name}: FETCH_SIZE {f:.0f} KiB (x2 = {2 * f / 1e6:.3f} GB), WRITE_SIZE {w:.0f} KiB ({w * 1024 / 1e9:.3f} GB), total {(2 * f + w) * 1024 / 1e9:.3f} GB per launch")
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', os.path.basename(out.rstrip('/')), 'traffic.json'), 'w'), indent=1)
