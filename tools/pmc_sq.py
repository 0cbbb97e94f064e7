"""SQ counter summaries of the dominant kernels -> profiles/*_pmc_sq_<tag>.txt and the MFMA-busy fraction bench.py quotes beside roofline.frac.
Usage: python tools/pmc_sq.py OUTDIR TAG   (expects OUTDIR/pmc_sq_<kernel tag>/summary.txt written by tools/pmc_kernel.sh)

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the busy counter sums, over all SIMDs, the cycles their matrix
pipe is occupied (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.  Entries are
keyed on the hash of the kernel sources, like profiles/traffic.json."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
out, tag = sys.argv[1], sys.argv[2]
names = {'proj_train': 'nr_qkv_proj_fwd[S=20]', 'attn_fwd': 'nr_attn_fwd[S=20]', 'attn_pool_fwd': 'nr_attn_pool_fwd[S=20]', 'attn_bwd_hm': 'nr_attn_bwd[S=20]', 'additive_bwd': 'nr_additive_bwd[S=20] (sequence-shaped)', 'pool_flat': 'nr_additive_bwd[S=20]', 'pool_flat50_act': 'nr_additive_bwd[abstract]',
         'additive_fwd': 'nr_additive_fwd[S=20]', 'additive_bwd50': 'nr_additive_bwd[abstract] (sequence-shaped)', 'conv_abs': 'nr_conv3_fwd[abstract]',
         'dx_gemm': 'nr_dx_gemm[S=20]', 'tn_gemm': 'nr_gemm_tn_dWqkv[S=20]', 'pool_fwd_flat50': 'nr_additive_fwd[abstract]', 'cgemm_dgrad50': 'nr_conv3_dgrad[abstract]'}
res = {"_note": __doc__.split('\n', 2)[2].strip()}
for k, name in names.items():
    f = os.path.join(out, f'pmc_sq_{k}', 'summary.txt')
    if not os.path.exists(f):
        continue
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r'^(\S+)\s+n=\s*\d+\s+avg=(\S+)', open(f).read(), re.M)}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c.get('GRBM_GUI_ACTIVE'):
        frac = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0)
        wave = c.get('SQ_WAVE_CYCLES', 0)
        res[name] = {"mfma_busy_frac": frac, "wait_any_frac": c.get('SQ_WAIT_ANY', 0) / wave if wave else None,
                     "wait_inst_frac": c.get('SQ_WAIT_INST_ANY', 0) / wave if wave else None,
                     "valu_per_mfma": c.get('SQ_INSTS_VALU', 0) / c['SQ_INSTS_MFMA'] if c.get('SQ_INSTS_MFMA') else None,
                     "lds_bank_conflict_frac": c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE'] if c.get('SQ_LDS_IDX_ACTIVE') else None,
                     "source": f"profiles/{tag}_pmc_sq_{k}.txt", "source_hash": bench.kernel_source_hash(), "workload": "NRMS/small/B512"}
        print(name, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in res[name].items() if a not in ('source', 'source_hash', 'workload')})
json.dump(res, open(os.path.join(out, 'mfma_busy.json'), 'w'), indent=1)
