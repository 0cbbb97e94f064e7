// libnr_engine.so -- C-ABI entry points (include/nr_engine.h) and kernel launches.
#include "nr_common.h"
#include "k_misc.h"
#include "k_mhsa_fwd.h"
#include "k_additive_fwd.h"
#include "k_bwd.h"
#include "k_proj.h"
#include "k_attn_bwd2.h"
#include "k_gemm.h"
#include "k_conv.h"
#include "k_convgemm.h"
#include "k_pool3.h"
#include "k_pool4.h"
#include "k_naml.h"
#include "k_gru.h"
#ifndef NR_EMU      // inter-workgroup waits: nothing the emulator (one workgroup after the other) can run
#include "k_xcd.h"
#include "k_gru_persist.h"
#endif
#include "k_eval.h"
#include "k_optim.h"
#include "k_sort.h"
#include "k_generic.h"
#include "k_step.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

namespace nr {
// defined in nr_mhsa2.hip: its own translation unit because the register-resident kernel wants the AGPR half of the register
// file as storage (default MFMA form), while every other kernel is built with -amdgpu-mfma-vgpr-form
int launch_mhsa_fwd2(const MhsaParams& p, hipStream_t stream);
int launch_pool2_bwd(const AdditiveBwdParams& p, hipStream_t stream);   // k_pool2.h, same translation unit
int launch_pool2_bwd50(const AdditiveBwdParams& p, hipStream_t stream);
}

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int fail(int code, const char* who, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s%s", who, msg);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return NR_ERR_LAUNCH;
  }
  return NR_OK;
}

// device-resident step counter of the process (nr_set_step_counter); null: seeds and step indices are taken by value as given
const uint32_t* g_step_ctr = nullptr;

nr::DropCfg make_drop(float p, uint64_t seed) {
  nr::DropCfg dc;
  dc.ctr = g_step_ctr;
  dc.enabled = p > 0.0f ? 1 : 0;
  dc.k0 = (uint32_t)seed;
  dc.k1 = (uint32_t)(seed >> 32);
  double t = (double)p * 65536.0 + 0.5;
  dc.thresh = t >= 65535.0 ? 65535u : (uint32_t)t;
  dc.scale = p > 0.0f ? 1.0f / (1.0f - p) : 1.0f;
  return dc;
}

int grid_for(int64_t work, int per_block, int cap) {
  int64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// 50-token sequences (NAML abstracts, click histories): the register-resident backward of k_pool2.h (<50, 1, 4>: 4 sequences per
// workgroup) from 2048 sequences up -- measured on MI355X: 27,136 abstracts 1.31 -> 1.03 ms, but 512 histories 36 -> 45 us (128 workgroups
// leave half of the CUs idle), so short batches keep the LDS-tile kernel (one sequence per workgroup).  NR_POOL2_S50=0: always the
// LDS-tile kernel, 2: always the register-resident one.
bool pool2_s50(int64_t n_seq) {
  static int v = -1;
  if (v < 0) { const char* e = getenv("NR_POOL2_S50"); v = e ? atoi(e) : 1; }
  return v == 2 || (v == 1 && n_seq >= 2048);
}

template <typename K>
int allow_smem(K kern, int bytes) { return nr::set_max_dynamic_lds((const void*)kern, bytes); }   // > 64 KiB needs the opt-in

// ---- general ring GEMMs (csrc/k_gemm.h) --------------------------------------------------------------------------------------------
template <int MODE, int WR, int TM, int TN_>
static int launch_gemm(const nr::GemmParams& p, int64_t grid, void* stream, const char* who) {
  using G = nr::GemmGeom<MODE, WR, TM, TN_>;
  if (allow_smem(nr::gemm_ring_kernel<MODE, WR, TM, TN_>, G::SMEM)) return fail(NR_ERR_LAUNCH, who, ": cannot reserve LDS");
  NR_LAUNCH((nr::gemm_ring_kernel<MODE, WR, TM, TN_>), grid, 512, G::SMEM, (hipStream_t)stream, p);
  return check_launch(who);
}

template <int S, int NSEQ, int NW>
int launch_conv_t(nr::ConvParams& p, void* stream) {
  using G = nr::ConvGeom<S, NSEQ>;
  if (allow_smem(nr::conv3_kernel<S, NSEQ, NW>, G::SMEM)) return fail(NR_ERR_LAUNCH, "conv3: cannot reserve LDS");
  NR_LAUNCH((nr::conv3_kernel<S, NSEQ, NW>), (p.n_seq + NSEQ - 1) / NSEQ, NW * 64, G::SMEM, (hipStream_t)stream, p);
  return NR_OK;
}

// The persistent form of the convolution GEMM (csrc/k_convgemm.h): usable while the token rows fit a 2 GiB buffer resource.  NR_CONV_GEMM_PERSIST=0
// (A/B switch, read once) keeps the one-tile-per-workgroup kernels.
static bool conv_gemm_persist_ok(int64_t n_seq, int S) {
  static const int on = [] { const char* e = getenv("NR_CONV_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
  return on != 0 && n_seq * S * (int64_t)(NR_KP * 2) < (1LL << 31) && n_seq * (S + 1) < (1LL << 31);
}
// Form of nr_dx_gemm AND of its packed operand (nr_pack_qkv_dx / nr_pack_encoder write what the kernel of this process reads): 0 (default) the
// one-tile-per-workgroup ring of k_proj.h over a fragment-ordered operand, NR_DX_STREAM=1 the persistent stream kernel of k_convgemm.h over a
// row-major operand -- a TIE on MI355X (365 - 369 vs 367 - 371 us per NRMS step, profiles/r06_ab_dx_stream.txt; with the conv form's tap-inner
// chunk order it LOST, 389 - 393 us: the two 64-byte halves of a 128-byte line were requested three chunks apart).  Read once.
static int dx_stream_form() {
  static const int on = [] { const char* e = getenv("NR_DX_STREAM"); return e ? (atoi(e) != 0 ? 1 : 0) : 0; }();
  return on;
}
static int launch_conv_gemm(nr::ConvGemmParams& p, int64_t n_seq, int S, void* stream, const char* who) {
  using G = nr::ConvGemmGeom;
  p.n_rows = n_seq * (S + 1) - 1; p.n_tok = n_seq * S;
  p.n_tiles = (int)((p.n_rows + G::BN - 1) / G::BN);
  p.S1 = (uint32_t)(S + 1); p.s1_magic = (uint32_t)((1ULL << 32) / (uint32_t)(S + 1)) + 1u;
  { const char* d = getenv("NR_CONVGEMM_DEBUG"); p.debug = d ? atoi(d) : 0; }      // profiling: phase switches (re-read per call)
  const int cus = nr::device_cus();
  if (p.debug) {
    if (allow_smem(nr::conv_gemm_kernel<true>, G::SMEM)) return fail(NR_ERR_LAUNCH, who, ": cannot reserve LDS");
    NR_LAUNCH((nr::conv_gemm_kernel<true>), p.n_tiles < cus ? p.n_tiles : cus, 512, G::SMEM, (hipStream_t)stream, p);
    return check_launch(who);
  }
  // chunk order of a tile (k_convgemm.h): pairs of column blocks under each tap (default) or tap-inner (NR_CONVGEMM_PAIRS=0); read once.
  // NAML step, one box: titles 316 / 318 vs 325 / 322 us, abstracts 748 / 763 vs 764 / 760 us (profiles/r06_ab_convgemm_pairs.txt)
  static const int pairs = [] { const char* e = getenv("NR_CONVGEMM_PAIRS"); return e ? atoi(e) : 1; }();
  if (pairs) {
    if (allow_smem(nr::conv_gemm_kernel<false, false, true>, G::SMEM)) return fail(NR_ERR_LAUNCH, who, ": cannot reserve LDS");
    NR_LAUNCH((nr::conv_gemm_kernel<false, false, true>), p.n_tiles < cus ? p.n_tiles : cus, 512, G::SMEM, (hipStream_t)stream, p);
    return check_launch(who);
  }
  if (allow_smem(nr::conv_gemm_kernel<false>, G::SMEM)) return fail(NR_ERR_LAUNCH, who, ": cannot reserve LDS");
  NR_LAUNCH((nr::conv_gemm_kernel<false>), p.n_tiles < cus ? p.n_tiles : cus, 512, G::SMEM, (hipStream_t)stream, p);
  return check_launch(who);
}

}  // namespace

extern "C" {

int nr_version(void) { return 1; }
const char* nr_last_error(void) { return g_err; }
int nr_supported_seq_len(int S) { return (S == 20 || S == 50) ? 1 : 0; }

int nr_gather_rows_f32(const int64_t* ids, const float* table, float* out, int64_t n_tokens, int d, int64_t num_rows,
                       void* stream) {
  if (!ids || !table || !out || d <= 0 || (d & 3) || num_rows <= 0 || n_tokens < 0) return fail(NR_ERR_BADARG, "nr_gather_rows_f32: bad argument");
  if (n_tokens == 0) return NR_OK;
  int grid = grid_for(n_tokens * (d / 4), 256, 256 * 8);
  NR_LAUNCH(nr::gather_rows_kernel, grid, 256, 0, (hipStream_t)stream, ids, table, out, n_tokens, d / 4, num_rows);
  return check_launch("nr_gather_rows_f32");
}

int nr_pack_qkv(const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                uint16_t* Wp, float* bp, void* stream) {
  if (!Wq || !bq || !Wk || !bk || !Wv || !bv || !Wp || !bp) return fail(NR_ERR_BADARG, "nr_pack_qkv: null pointer");
  NR_LAUNCH(nr::pack_qkv_kernel, 256, 256, 0, (hipStream_t)stream, Wq, bq, Wk, bk, Wv, bv, Wp, bp);
  return check_launch("nr_pack_qkv");
}

int nr_pack_additive(const float* Wa, const float* ba, const float* qv, int qdim, uint16_t* Wap, float* bap, float* qvp,
                     void* stream) {
  if (!Wa || !ba || !qv || !Wap || !bap || !qvp) return fail(NR_ERR_BADARG, "nr_pack_additive: null pointer");
  if (qdim <= 0 || qdim > NR_QP) return fail(NR_ERR_UNSUPPORTED, "nr_pack_additive: query_vector_dim must be in [1,208]");
  NR_LAUNCH(nr::pack_additive_kernel, 64, 256, 0, (hipStream_t)stream, Wa, ba, qv, qdim, Wap, bap, qvp);
  return check_launch("nr_pack_additive");
}

int nr_wgrad_unpack(const float* dW_parts, int nc_w, const float* dWa_parts, int nc_a, const float* dq_part, int64_t nwg, int qdim,
                    float* gWq, float* gbq, float* gWk, float* gbk, float* gWv, float* gbv, float* gWa, float* gba, float* gq, void* stream) {
  if (!dW_parts || !dWa_parts || !dq_part || !gWq || !gbq || !gWk || !gbk || !gWv || !gbv || !gWa || !gba || !gq)
    return fail(NR_ERR_BADARG, "nr_wgrad_unpack: null pointer");
  if (nc_w <= 0 || nc_a <= 0 || nwg < 0) return fail(NR_ERR_BADARG, "nr_wgrad_unpack: bad partial counts");
  if (qdim <= 0 || qdim > NR_QP) return fail(NR_ERR_UNSUPPORTED, "nr_wgrad_unpack: query_vector_dim must be in [1,208]");
  nr::WgradUnpackParams p;
  p.dW = dW_parts; p.dWa = dWa_parts; p.dq = dq_part; p.ncW = nc_w; p.ncA = nc_a; p.qdim = qdim; p.nwg = nwg;
  p.gW[0] = gWq; p.gW[1] = gWk; p.gW[2] = gWv; p.gb[0] = gbq; p.gb[1] = gbk; p.gb[2] = gbv; p.gWa = gWa; p.gba = gba; p.gq = gq;
  const uintptr_t al = (uintptr_t)gWq | (uintptr_t)gWk | (uintptr_t)gWv | (uintptr_t)gWa | (uintptr_t)dW_parts | (uintptr_t)dWa_parts;
  if ((al & 15) == 0) NR_LAUNCH(nr::wgrad_unpack_kernel<true>, nr::wgrad_unpack_grid(qdim), nr::WGU_THREADS, nr::WGU_SMEM, (hipStream_t)stream, p);
  else if ((((uintptr_t)dW_parts | (uintptr_t)dWa_parts) & 15) == 0)
    NR_LAUNCH(nr::wgrad_unpack_kernel<false>, nr::wgrad_unpack_grid(qdim), nr::WGU_THREADS, nr::WGU_SMEM, (hipStream_t)stream, p);
  else return fail(NR_ERR_BADARG, "nr_wgrad_unpack: the partial-product buffers must be 16-byte aligned");
  NR_LAUNCH(nr::wgrad_dq_kernel, (qdim + nr::WGU_DQ_COLS - 1) / nr::WGU_DQ_COLS, nr::WGU_DQ_COLS * nr::WGU_DQ_PH,
            nr::WGU_DQ_COLS * nr::WGU_DQ_PH * 4, (hipStream_t)stream, p);
  return check_launch("nr_wgrad_unpack");
}

int nr_mhsa_fwd_len(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense, const uint16_t* Wp,
                    const float* bp, uint16_t* ctx, uint16_t* q_save, uint16_t* k_save, uint16_t* vt_save, uint16_t* x_save, const int32_t* key_len,
                    int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream);

int nr_mhsa_fwd_ex(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense, const uint16_t* Wp,
                   const float* bp, uint16_t* ctx, uint16_t* q_save, uint16_t* k_save, uint16_t* vt_save, uint16_t* x_save, int64_t n_seq,
                   int S, float p_drop, uint64_t seed, void* stream) {
  return nr_mhsa_fwd_len(ids, table, num_rows, x_dense, Wp, bp, ctx, q_save, k_save, vt_save, x_save, nullptr, n_seq, S, p_drop, seed, stream);
}

int nr_mhsa_fwd(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense, const uint16_t* Wp,
                const float* bp, uint16_t* ctx, uint16_t* q_save, uint16_t* k_save, uint16_t* vt_save, int64_t n_seq, int S,
                float p_drop, uint64_t seed, void* stream) {
  return nr_mhsa_fwd_ex(ids, table, num_rows, x_dense, Wp, bp, ctx, q_save, k_save, vt_save, nullptr, n_seq, S, p_drop, seed, stream);
}

int nr_mhsa_fwd_len(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense, const uint16_t* Wp,
                    const float* bp, uint16_t* ctx, uint16_t* q_save, uint16_t* k_save, uint16_t* vt_save, uint16_t* x_save, const int32_t* key_len,
                    int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream) {
  if (!Wp || !bp || !ctx || n_seq < 0) return fail(NR_ERR_BADARG, "nr_mhsa_fwd: bad argument");
  if ((q_save != nullptr) != (k_save != nullptr) || (q_save != nullptr) != (vt_save != nullptr))
    return fail(NR_ERR_BADARG, "nr_mhsa_fwd: q_save / k_save / vt_save must be given together");
  if ((ids == nullptr) == (x_dense == nullptr)) return fail(NR_ERR_BADARG, "nr_mhsa_fwd: exactly one of ids / x_dense");
  if (ids && (!table || num_rows <= 0)) return fail(NR_ERR_BADARG, "nr_mhsa_fwd: ids without table");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_mhsa_fwd: dropout probability out of range");
  if (n_seq == 0) return NR_OK;
  nr::MhsaParams p;
  p.ids = ids; p.table = table; p.num_rows = num_rows; p.x_dense = x_dense;
  p.Wp = Wp; p.bp = bp; p.ctx = ctx; p.n_seq = n_seq; p.key_len = key_len; p.dc = make_drop(p_drop, seed); p.debug = 0;
  p.q_save = q_save; p.k_save = k_save; p.vt_save = vt_save; p.x_save = nullptr;
  bool x_done = false;
  if (S == 20) {
    // the register-resident kernel (csrc/nr_mhsa2.hip); the LDS-tile kernel remains for 50-token sequences below
    p.x_save = x_save; x_done = true;
    if (nr::launch_mhsa_fwd2(p, (hipStream_t)stream)) return fail(NR_ERR_LAUNCH, "nr_mhsa_fwd: cannot reserve LDS");
  } else if (S == 50) {
    constexpr int NSEQ = 1, NW = 4, GS = 2;
    using G = nr::MhsaGeom<50, NSEQ, NW>;
    if (allow_smem(nr::mhsa_fwd_kernel<50, NSEQ, NW, GS>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_mhsa_fwd: cannot reserve LDS");
    NR_LAUNCH((nr::mhsa_fwd_kernel<50, NSEQ, NW, GS>), (n_seq + NSEQ - 1) / NSEQ, G::THREADS, G::SMEM, (hipStream_t)stream, p);
  } else {
    return fail(NR_ERR_UNSUPPORTED, "nr_mhsa_fwd: sequence length not instantiated (20, 50)");
  }
  if (x_save != nullptr && !x_done)        // kernels that do not emit the token matrix themselves: a separate gather pass
    NR_LAUNCH(nr::gather_bf16_kernel, grid_for(n_seq * S * (NR_KP / 4), 256, 4096), 256, 0, (hipStream_t)stream, ids, table, num_rows, x_dense,
              x_save, n_seq * S, p.dc);
  return check_launch("nr_mhsa_fwd");
}

int nr_supported_pool_len(int S) { return (S == 4 || S == 20 || S == 50) ? 1 : 0; }
int nr_supported_conv_len(int S) { return (S == 20 || S == 50) ? 1 : 0; }

int nr_additive_fwd_v(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out, int64_t out_stride,
                      uint16_t* out_b, int64_t out_b_stride, float* attn_w, int64_t n_seq, int S, int valid, void* stream);

int nr_additive_fwd_ex(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out, int64_t out_stride,
                       uint16_t* out_b, int64_t out_b_stride, float* attn_w, int64_t n_seq, int S, void* stream) {
  return nr_additive_fwd_v(ctx, Wap, bap, qvp, out, out_stride, out_b, out_b_stride, attn_w, n_seq, S, S, stream);
}

int nr_additive_fwd_v(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out, int64_t out_stride,
                      uint16_t* out_b, int64_t out_b_stride, float* attn_w, int64_t n_seq, int S, int valid, void* stream) {
  if (!ctx || !Wap || !bap || !qvp || (!out && !out_b) || n_seq < 0 || valid < 1 || valid > S) return fail(NR_ERR_BADARG, "nr_additive_fwd: bad argument");
  if ((out && (out_stride < NR_D || (out_stride & 3))) || (out_b && (out_b_stride < NR_KP || (out_b_stride & 7))))
    return fail(NR_ERR_BADARG, "nr_additive_fwd: bad output stride");
  if (n_seq == 0) return NR_OK;
  nr::AdditiveParams p;
  p.ctx = ctx; p.Wap = Wap; p.bap = bap; p.qvp = qvp; p.out = out; p.out_stride = out_stride; p.out_b = out_b;
  p.out_b_stride = out_b_stride; p.attn_w = attn_w; p.n_seq = n_seq; p.valid = valid;
  if (S == 20) {                      // 2 titles per workgroup (5 workgroups per CU hide the per-tile latency chain: -14 % against 4; 8 waves on 8 titles lost too:
    constexpr int NSEQ = 2;           //  profiles/r02_ab_switches.txt -- those variants are gone)
    using G = nr::AddGeom<20, NSEQ>;
    if (allow_smem(nr::additive_fwd_kernel<20, NSEQ>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_fwd: cannot reserve LDS");
    NR_LAUNCH((nr::additive_fwd_kernel<20, NSEQ>), (n_seq + NSEQ - 1) / NSEQ, nr::WG, G::SMEM, (hipStream_t)stream, p);
  } else if (S == 50) {
    constexpr int NSEQ = 1;          // (2 sequences per workgroup measured no faster for 27k abstracts and 2x slower for 512 histories)
    using G = nr::AddGeom<50, NSEQ>;
    if (allow_smem(nr::additive_fwd_kernel<50, NSEQ>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_fwd: cannot reserve LDS");
    NR_LAUNCH((nr::additive_fwd_kernel<50, NSEQ>), (n_seq + NSEQ - 1) / NSEQ, nr::WG, G::SMEM, (hipStream_t)stream, p);
  } else if (S == 4) {
    constexpr int NSEQ = 20;
    using G = nr::AddGeom<4, NSEQ>;
    if (allow_smem(nr::additive_fwd_kernel<4, NSEQ>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_fwd: cannot reserve LDS");
    NR_LAUNCH((nr::additive_fwd_kernel<4, NSEQ>), (n_seq + NSEQ - 1) / NSEQ, nr::WG, G::SMEM, (hipStream_t)stream, p);
  } else {
    return fail(NR_ERR_UNSUPPORTED, "nr_additive_fwd: sequence length not instantiated (4, 20, 50)");
  }
  return check_launch("nr_additive_fwd");
}

// The same forward over whole sequences per wave, persistent (csrc/k_pool4.h): any S in [16, 64], query_vector_dim <= 200 (the rows of Wa the kernel
// keeps in LDS).  Same outputs as nr_additive_fwd_v up to the order of fp32 additions.
int nr_additive_fwd_flat(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out, int64_t out_stride, uint16_t* out_b,
                         int64_t out_b_stride, float* attn_w, int64_t n_seq, int S, int valid, int qdim, void* stream) {
  if (!ctx || !Wap || !bap || !qvp || (!out && !out_b) || n_seq < 0 || valid < 1 || valid > S) return fail(NR_ERR_BADARG, "nr_additive_fwd_flat: bad argument");
  if ((out && (out_stride < NR_D || (out_stride & 3))) || (out_b && (out_b_stride < NR_KP || (out_b_stride & 7))))
    return fail(NR_ERR_BADARG, "nr_additive_fwd_flat: bad output stride");
  if (S < 16 || S > nr::Pool4Geom::ROWS || qdim < 1 || qdim > nr::Pool4Geom::WROWS || n_seq * S >= (1LL << 31))
    return fail(NR_ERR_UNSUPPORTED, "nr_additive_fwd_flat: needs 16 <= S <= 64 and query_vector_dim <= 200");
  if (n_seq == 0) return NR_OK;
  nr::Pool4Params p;
  p.ctx = ctx; p.Wap = Wap; p.bap = bap; p.qvp = qvp; p.out = out; p.out_stride = out_stride; p.out_b = out_b; p.out_b_stride = out_b_stride;
  p.attn_w = attn_w; p.n_seq = n_seq; p.S = S; p.valid = valid;
  const int nslot = nr::Pool4Geom::ROWS / S;
  const int64_t groups = (n_seq + nslot - 1) / nslot, wgs = (groups + nr::Pool4Geom::NWAVE - 1) / nr::Pool4Geom::NWAVE;
  const int cus = nr::device_cus();
  const char* d = getenv("NR_POOL_DEBUG");        // profiling: phase switches of the DBG instantiation (re-read per call; tools/prof_kernel.py)
  p.dbg = d ? atoi(d) : 0;
  if (p.dbg) {
    if (allow_smem(nr::pool4_fwd_kernel<true>, nr::Pool4Geom::SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_fwd_flat: cannot reserve LDS");
    NR_LAUNCH(nr::pool4_fwd_kernel<true>, wgs < cus ? wgs : cus, nr::Pool4Geom::THREADS, nr::Pool4Geom::SMEM, (hipStream_t)stream, p);
    return check_launch("nr_additive_fwd_flat");
  }
  if (allow_smem(nr::pool4_fwd_kernel<false>, nr::Pool4Geom::SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_fwd_flat: cannot reserve LDS");
  NR_LAUNCH(nr::pool4_fwd_kernel<false>, wgs < cus ? wgs : cus, nr::Pool4Geom::THREADS, nr::Pool4Geom::SMEM, (hipStream_t)stream, p);
  return check_launch("nr_additive_fwd_flat");
}

int nr_additive_fwd(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, float* out,
                    float* attn_w, int64_t n_seq, int S, void* stream) {
  if (!out) return fail(NR_ERR_BADARG, "nr_additive_fwd: bad argument");
  return nr_additive_fwd_ex(ctx, Wap, bap, qvp, out, NR_D, nullptr, 0, attn_w, n_seq, S, stream);
}

int nr_attn_bwd_len(const uint16_t* q_save, const uint16_t* k_save, const uint16_t* vt_save, const uint16_t* dctx_gemm, int ldc,
                    const float* attn_w, const float* g_out, uint16_t* dqkv, const int32_t* key_len, int64_t n_seq, int S, float p_drop,
                    uint64_t seed, void* stream);

int nr_attn_bwd(const uint16_t* q_save, const uint16_t* k_save, const uint16_t* vt_save, const uint16_t* dctx_gemm, int ldc,
                const float* attn_w, const float* g_out, uint16_t* dqkv, int64_t n_seq, int S, float p_drop, uint64_t seed,
                void* stream) {
  return nr_attn_bwd_len(q_save, k_save, vt_save, dctx_gemm, ldc, attn_w, g_out, dqkv, nullptr, n_seq, S, p_drop, seed, stream);
}

static unsigned long long* g_attnb_stamps = nullptr;
int nr_debug_attnb_stamps(uint64_t* buf) { g_attnb_stamps = (unsigned long long*)buf; return NR_OK; }

static int attn_bwd_launch(const uint16_t* q_save, const uint16_t* k_save, const uint16_t* vt_save, int hm, const uint16_t* dctx_gemm, int ldc,
                           const float* attn_w, const float* g_out, uint16_t* dqkv, const int32_t* key_len, int64_t n_seq, int S, float p_drop,
                           uint64_t seed, void* stream) {
  if (!q_save || (!hm && (!k_save || !vt_save)) || !dctx_gemm || !attn_w || !g_out || !dqkv || n_seq < 0 || ldc < NR_D || (ldc & 3))
    return fail(NR_ERR_BADARG, "nr_attn_bwd: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_attn_bwd: dropout probability out of range");
  if (hm && S != 20) return fail(NR_ERR_UNSUPPORTED, "nr_attn_bwd_hm: the head-major layout is instantiated for 20-token sequences");
  if (n_seq == 0) return NR_OK;
  nr::AttnBwdParams p;
  p.q_save = q_save; p.k_save = k_save; p.vt_save = vt_save; p.dctx_gemm = dctx_gemm; p.ldc = ldc; p.attn_w = attn_w;
  p.g_out = g_out; p.dqkv = dqkv; p.n_seq = n_seq; p.key_len = key_len; p.dc = make_drop(p_drop, seed); p.hm = hm; p.debug = 0;
  p.stamps = g_attnb_stamps;
  // head-major saves: a pair's operands are contiguous, nothing is shared between the heads of a token row except the dqkv row that is written
  p.xcd_major = 1;
  const int64_t pairs = n_seq * NR_HEADS;
  // persistent grid: each wave walks pairs with a stride and prefetches the next one.  NR_ATTN_BWD_MAX_WGS caps the
  // grid (used by the tests to force many pairs per wave on small inputs).
  const char* capenv = getenv("NR_ATTN_BWD_MAX_WGS");      // (re-read per call: the tests flip it inside one process)
  const int capdiv = capenv ? atoi(capenv) : 0;
  // head-major saves, contiguous 16-byte aligned operands: the DMA form (csrc/k_attn_bwd2.h).  NR_ATTNB2=0: the round-4 TILE kernel (A/B only)
  // (2: required -- the call fails instead of falling back; re-read per call: the tests flip it inside one process)
  const char* v2env = getenv("NR_ATTNB2");
  const int v2 = v2env ? atoi(v2env) : 1;
  const bool v2ok = hm && S == 20 && ldc == NR_KP &&
      ((((uintptr_t)q_save) | ((uintptr_t)dctx_gemm) | ((uintptr_t)attn_w) | ((uintptr_t)g_out) | ((uintptr_t)dqkv)) & 15) == 0;
  if (v2 == 2 && !v2ok) return fail(NR_ERR_UNSUPPORTED, "nr_attn_bwd: NR_ATTNB2=2 needs head-major saves, ldc == 320 and 16-byte aligned operands");
  if (v2 && v2ok) {
    nr::AttnBwd2Params q;
    q.qkv = q_save; q.dctx = dctx_gemm; q.attn_w = attn_w; q.g_out = g_out; q.dqkv = dqkv; q.n_seq = n_seq; q.key_len = key_len; q.dc = p.dc;
    q.debug = 0; q.stamps = g_attnb_stamps;
    const int64_t cap = capdiv > 0 ? capdiv : 2 * (int64_t)nr::device_cus();        // persistent: two workgroups per CU walk the titles
    const int grid = (int)(n_seq < cap ? n_seq : cap);
    const char* d = getenv("NR_ATTNB_DEBUG");       // profiling: phase switches of the DBG instantiation, re-read per call
    q.debug = d ? atoi(d) : 0;
#define NR_AB2_LAUNCH(NW_, DBG_)                                                                                                  \
    do {                                                                                                                          \
      using G2 = nr::AttnBwd2Geom<NW_>;                                                                                           \
      if (allow_smem(nr::attn_bwd2_kernel<NW_, DBG_>, G2::SMEM)) return fail(NR_ERR_LAUNCH, "nr_attn_bwd: cannot reserve LDS");  \
      NR_LAUNCH((nr::attn_bwd2_kernel<NW_, DBG_>), grid, G2::NT, G2::SMEM, (hipStream_t)stream, q);                              \
    } while (0)
    // eight waves: two rounds of heads, four waves per SIMD (five waves / three rounds measured 523 - 540 us against 487: profiles/r05_ab_notes.txt)
    if (q.debug) NR_AB2_LAUNCH(8, true); else NR_AB2_LAUNCH(8, false);
#undef NR_AB2_LAUNCH
    return check_launch("nr_attn_bwd");
  }
  if (S == 20) {
    // row-major saves (and NR_ATTNB2=0): one workgroup per sequence, dqkv rows staged in LDS (the round-4 TILE kernel, csrc/k_bwd.h)
    const char* d = getenv("NR_ATTNB_DEBUG");       // profiling: phase switches (AttnBwdParams::debug), re-read per call
    constexpr int TW = 4;                           // 15 heads = 4 rounds of 4 waves (the last round: 3): two workgroups per CU at 2 waves per SIMD
    using GT = nr::AttnBwdGeom<20, TW>;
    const int grid = (int)(n_seq < (capdiv > 0 ? capdiv : 256 * 8) ? n_seq : (capdiv > 0 ? capdiv : 256 * 8));
    if (d != nullptr && atoi(d) != 0) {
      p.debug = atoi(d);
      if (allow_smem(nr::attn_bwd_kernel<20, TW, true, true>, GT::SMEM_TILE)) return fail(NR_ERR_LAUNCH, "nr_attn_bwd: cannot reserve LDS");
      NR_LAUNCH((nr::attn_bwd_kernel<20, TW, true, true>), grid, TW * 64, GT::SMEM_TILE, (hipStream_t)stream, p);
    } else {
      if (allow_smem(nr::attn_bwd_kernel<20, TW, false, true>, GT::SMEM_TILE)) return fail(NR_ERR_LAUNCH, "nr_attn_bwd: cannot reserve LDS");
      NR_LAUNCH((nr::attn_bwd_kernel<20, TW, false, true>), grid, TW * 64, GT::SMEM_TILE, (hipStream_t)stream, p);
    }
  } else if (S == 50) {
    constexpr int WPB = 2;
    using G = nr::AttnBwdGeom<50, WPB>;
    if (allow_smem(nr::attn_bwd_kernel<50, WPB>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_attn_bwd: cannot reserve LDS");
    NR_LAUNCH((nr::attn_bwd_kernel<50, WPB>), grid_for(pairs, WPB, capdiv > 0 ? capdiv : 256 * 8), WPB * 64, G::SMEM, (hipStream_t)stream, p);
  } else {
    return fail(NR_ERR_UNSUPPORTED, "nr_attn_bwd: sequence length not instantiated (20, 50)");
  }
  return check_launch("nr_attn_bwd");
}

int nr_attn_bwd_len(const uint16_t* q_save, const uint16_t* k_save, const uint16_t* vt_save, const uint16_t* dctx_gemm, int ldc,
                    const float* attn_w, const float* g_out, uint16_t* dqkv, const int32_t* key_len, int64_t n_seq, int S, float p_drop,
                    uint64_t seed, void* stream) {
  return attn_bwd_launch(q_save, k_save, vt_save, 0, dctx_gemm, ldc, attn_w, g_out, dqkv, key_len, n_seq, S, p_drop, seed, stream);
}

int nr_attn_bwd_hm(const uint16_t* qkv, const uint16_t* dctx_gemm, int ldc, const float* attn_w, const float* g_out, uint16_t* dqkv,
                   const int32_t* key_len, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream) {
  return attn_bwd_launch(qkv, nullptr, nullptr, 1, dctx_gemm, ldc, attn_w, g_out, dqkv, key_len, n_seq, S, p_drop, seed, stream);
}

int nr_pack_qkv32(const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv,
                  uint16_t* Wp32, float* bp, void* stream) {
  if (!Wq || !bq || !Wk || !bk || !Wv || !bv || !Wp32 || !bp) return fail(NR_ERR_BADARG, "nr_pack_qkv32: null pointer");
  NR_LAUNCH(nr::pack_qkv32_kernel, 256, 256, 0, (hipStream_t)stream, Wq, bq, Wk, bk, Wv, bv, Wp32, bp);
  return check_launch("nr_pack_qkv32");
}

int nr_pack_encoder(const float* Wq, const float* bq, const float* Wk, const float* bk, const float* Wv, const float* bv, const float* Wa,
                    const float* ba, const float* qv, int qdim, uint16_t* Wp, float* bp, uint16_t* Wp32, float* bp32, uint16_t* WdX,
                    uint16_t* Wap, float* bap, float* qvp, uint16_t* WaT, void* stream) {
  if (!Wq || !bq || !Wk || !bk || !Wv || !bv || !Wa || !ba || !qv) return fail(NR_ERR_BADARG, "nr_pack_encoder: null parameter");
  if ((Wp && !bp) || (Wp32 && !bp32) || (Wap && (!bap || !qvp))) return fail(NR_ERR_BADARG, "nr_pack_encoder: an operand without its bias / query vector");
  if (qdim <= 0 || qdim > NR_QP) return fail(NR_ERR_UNSUPPORTED, "nr_pack_encoder: query_vector_dim must be in [1,208]");
  nr::PackEncoderParams p{Wq, bq, Wk, bk, Wv, bv, Wa, ba, qv, qdim, Wp, bp, Wp32, bp32, WdX, dx_stream_form(), Wap, bap, qvp, WaT};
  NR_LAUNCH2(nr::pack_encoder_kernel, 128, 5, 256, 0, (hipStream_t)stream, p);
  return check_launch("nr_pack_encoder");
}

int nr_qkv_proj_fwd(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wp32, const float* bp, uint16_t* qkv,
                    uint16_t* x_save, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream) {
  if (!ids || !table || num_rows <= 0 || !Wp32 || !bp || !qkv || n_seq < 0) return fail(NR_ERR_BADARG, "nr_qkv_proj_fwd: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_qkv_proj_fwd: dropout probability out of range");
  if (S != 20) return fail(NR_ERR_UNSUPPORTED, "nr_qkv_proj_fwd: instantiated for 20-token sequences");
  if ((uint64_t)num_rows * (NR_D * 4) >= 0xFFFFFFF0ull || (uint64_t)n_seq * (NR_QKV_HM_SEQ * 2) >= (1ull << 40))
    return fail(NR_ERR_UNSUPPORTED, "nr_qkv_proj_fwd: the table is addressed with 32-bit byte offsets (at most 3,579,139 rows)");
  if (n_seq == 0) return NR_OK;
  nr::ProjParams p;
  p.ids = ids; p.table = table; p.num_rows = num_rows; p.Wp32 = Wp32; p.bp = bp; p.qkv = qkv; p.x_save = x_save; p.n_tok = n_seq * S;
  p.dc = make_drop(p_drop, seed); p.debug = 0;
  using G = nr::ProjGeom;
  const int64_t grid = (p.n_tok + G::TOK_WG - 1) / G::TOK_WG;
  const char* d = getenv("NR_PROJ_DEBUG");          // profiling: phase switches (ProjParams::debug), re-read per call
  if (allow_smem(nr::qkv_proj_kernel<true>, G::SMEM) || allow_smem(nr::qkv_proj_kernel<false>, G::SMEM))
    return fail(NR_ERR_LAUNCH, "nr_qkv_proj_fwd: cannot reserve LDS");
  if (d != nullptr && atoi(d) != 0) {
    p.debug = atoi(d);
    NR_LAUNCH((nr::qkv_proj_kernel<true>), grid, G::NWAVE * 64, G::SMEM, (hipStream_t)stream, p);
  } else {
    NR_LAUNCH((nr::qkv_proj_kernel<false>), grid, G::NWAVE * 64, G::SMEM, (hipStream_t)stream, p);
  }
  return check_launch("nr_qkv_proj_fwd");
}

int nr_pack_qkv_dx(const float* Wq, const float* Wk, const float* Wv, uint16_t* WdX, void* stream) {
  if (!Wq || !Wk || !Wv || !WdX) return fail(NR_ERR_BADARG, "nr_pack_qkv_dx: null pointer");
  NR_LAUNCH(nr::pack_qkv_dx_kernel, 256, 256, 0, (hipStream_t)stream, Wq, Wk, Wv, WdX, dx_stream_form());
  return check_launch("nr_pack_qkv_dx");
}

int nr_dx_gemm(const uint16_t* dqkv, const uint16_t* WdX, uint16_t* dX, int64_t n_tok, void* stream) {
  if (!dqkv || !WdX || !dX || n_tok < 0) return fail(NR_ERR_BADARG, "nr_dx_gemm: bad argument");
  if ((((uintptr_t)dqkv | (uintptr_t)WdX | (uintptr_t)dX) & 15) != 0) return fail(NR_ERR_BADARG, "nr_dx_gemm: buffers must be 16-byte aligned");
  if (n_tok == 0) return NR_OK;
  if (dx_stream_form()) {
    // the persistent stream kernel (k_convgemm.h, PLAIN): 32-bit output offsets -> launches of at most 2^31 / 640 output rows
    using G = nr::ConvGemmGeom;
    const int64_t seg = ((1LL << 31) / (NR_KP * 2) / G::BN - 1) * G::BN;
    const int cus = nr::device_cus();
    if (allow_smem(nr::conv_gemm_kernel<false, true>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_dx_gemm: cannot reserve LDS");
    for (int64_t t0 = 0; t0 < n_tok; t0 += seg) {
      nr::ConvGemmParams q{};
      q.A = WdX; q.R = dqkv + t0 * (3 * NR_KP); q.C = dX + t0 * NR_KP;
      q.n_rows = q.n_tok = n_tok - t0 < seg ? n_tok - t0 : seg;
      q.n_tiles = (int)((q.n_rows + G::BN - 1) / G::BN);
      NR_LAUNCH((nr::conv_gemm_kernel<false, true>), q.n_tiles < cus ? q.n_tiles : cus, 512, G::SMEM, (hipStream_t)stream, q);
    }
    return check_launch("nr_dx_gemm");
  }
  nr::DxParams p;
  p.dqkv = dqkv; p.WdX = WdX; p.dX = dX; p.n_tok = n_tok;
  using R = nr::DxRingGeom;
  const char* d = getenv("NR_DXR_DEBUG");      // profiling: phase switches (compile-time variants), re-read per call
  const int dbg = d != nullptr ? atoi(d) : 0;
  const int64_t grid = (n_tok + R::TOK_WG - 1) / R::TOK_WG;
#define NR_DXR_CASE(D) case D: if (allow_smem(nr::dx_gemm_ring_kernel<D>, R::SMEM)) return fail(NR_ERR_LAUNCH, "nr_dx_gemm: cannot reserve LDS"); \
                               NR_LAUNCH(nr::dx_gemm_ring_kernel<D>, grid, 512, R::SMEM, (hipStream_t)stream, p); break;
  switch (dbg) { NR_DXR_CASE(1) NR_DXR_CASE(2) NR_DXR_CASE(3) NR_DXR_CASE(4) NR_DXR_CASE(7) default: NR_DXR_CASE(0) }
#undef NR_DXR_CASE
  return check_launch("nr_dx_gemm");
}

// ---- general ring GEMMs (csrc/k_gemm.h; launch_gemm is defined above, outside the extern "C" block) -----------------------------------
int nr_gemm_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, void* stream) {
  if (!A || !B || !C || M < 0 || N <= 0 || K <= 0 || (K & 31) || lda < K || ldb < K || ldc < N || (lda & 7) || (ldb & 7))
    return fail(NR_ERR_BADARG, "nr_gemm_nt: bad argument (K must be a multiple of 32, row strides multiples of 8 elements)");
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return fail(NR_ERR_BADARG, "nr_gemm_nt: operands must be 16-byte aligned");
  if (M == 0) return NR_OK;
  nr::GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.tapw = 1 << 30;
  using G = nr::GemmGeom<0, 4, 2, 4>;
  p.tiles_m = (int)((M + G::BM - 1) / G::BM); p.tiles_n = (N + G::BN - 1) / G::BN;
  const int64_t grid = (int64_t)((p.tiles_m + 7) / 8) * 8 * p.tiles_n;
  return launch_gemm<0, 4, 2, 4>(p, grid, stream, "nr_gemm_nt");
}

// nr_gemm_nt whose A rows OVERLAP: row m reads the K contiguous elements from A + m * lda with lda < K allowed (a convolution window = w contiguous
// rows of a seqpad buffer: csrc/k_generic.h).  The caller guarantees (M - 1) * lda + K readable elements.
int nr_gemm_nt_rows(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, void* stream) {
  if (!A || !B || !C || M < 0 || N <= 0 || K <= 0 || (K & 31) || lda < 8 || ldb < K || ldc < N || (lda & 7) || (ldb & 7))
    return fail(NR_ERR_BADARG, "nr_gemm_nt_rows: bad argument (K must be a multiple of 32, row strides multiples of 8 elements)");
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return fail(NR_ERR_BADARG, "nr_gemm_nt_rows: operands must be 16-byte aligned");
  if (M == 0) return NR_OK;
  nr::GemmParams p{};
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.tapw = 1 << 30;
  using G = nr::GemmGeom<0, 4, 2, 4>;
  p.tiles_m = (int)((M + G::BM - 1) / G::BM); p.tiles_n = (N + G::BN - 1) / G::BN;
  const int64_t grid = (int64_t)((p.tiles_m + 7) / 8) * 8 * p.tiles_n;
  return launch_gemm<0, 4, 2, 4>(p, grid, stream, "nr_gemm_nt_rows");
}

// output tile of the TN kernel: 256 x 320 (4 x 2 waves of 2 x 5 tiles) when the output has at most 320 columns (projection gradients 960 x 320,
// pooling gradients 208 x 320), 320 x 256 (2 x 4 waves of 5 x 2 tiles) for 257 .. 320 rows (conv tap gradients: 320 filters x 3 x 320), else 256 x 256
static void gemm_tn_tile(int M, int N, int* bm, int* bn) {
  if (N > 256 && N <= 320) { *bm = 256; *bn = 320; return; }     // one column tile of 320: projection (960 x 320) and pooling (208 x 320) gradients
  *bm = (M > 256 && M <= 320) ? 320 : 256;                       // conv taps: 320 filters x (3 x 320)
  *bn = 256;
}

int nr_gemm_tn_parts(int M, int N, int64_t n_tok) {
  if (M <= 0 || N <= 0 || n_tok < 0) return -1;
  int BM, BN;
  gemm_tn_tile(M, N, &BM, &BN);
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int P = (256 + tiles - 1) / tiles;            // one workgroup per CU
  P = (P + 7) / 8 * 8;
  while (P > 8 && n_tok / P < 512) P -= 8;      // short token lists (the user encoder's 25,600 rows): at least 16 chunks per partition --
  if (tiles > 32) {                             // more output tiles than 8 partitions fill the device with evenly (the GRU's 2,736 x 928 weight gradients: 44 tiles)
    const char* e = getenv("NR_TN_P_MANY_TILES");        // A/B knob
    const int want = e ? atoi(e) : 8;
    if (want >= 8 && !(want & 7) && n_tok / want >= 512) P = want;
  }
  return P;                                     // every partition costs a full M x N block of fp32 partials

}

int nr_gemm_tn(const uint16_t* G, int64_t ldg, int M, const uint16_t* X, int64_t ldx, int tapw, int taps, const uint16_t* zeros, float* out,
               int64_t ldo, int64_t n_tok, int P, void* stream) {
  if (!G || !X || !zeros || !out || M <= 0 || M > ldg || (ldg & 7) || (ldx & 7) || tapw <= 0 || tapw > ldx || (tapw & 7) || taps < 1 || taps > 9 ||
      n_tok < 0 || P <= 0 || (P & 7) || ldo < (int64_t)taps * tapw)
    return fail(NR_ERR_BADARG, "nr_gemm_tn: bad argument");
  if ((((uintptr_t)G | (uintptr_t)X | (uintptr_t)zeros) & 15) != 0) return fail(NR_ERR_BADARG, "nr_gemm_tn: operands must be 16-byte aligned");
  nr::GemmParams p{};
  p.A = G; p.B = X; p.C = out; p.lda = ldg; p.ldb = ldx; p.ldc = ldo; p.M = M; p.N = taps * tapw; p.K = 0; p.zeros = zeros; p.n_tok = n_tok;
  p.P = P; p.tok_per_part = ((n_tok + P - 1) / P + 31) / 32 * 32; p.tapw = taps == 1 ? (1 << 30) : tapw;
  int BM, BN;
  gemm_tn_tile(M, p.N, &BM, &BN);
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  const int64_t grid = (int64_t)P * p.tiles_m * p.tiles_n;
  if (BM == 320) return launch_gemm<1, 2, 5, 2>(p, grid, stream, "nr_gemm_tn");
  if (BN == 320) return launch_gemm<1, 4, 2, 5>(p, grid, stream, "nr_gemm_tn");
  return launch_gemm<1, 4, 2, 4>(p, grid, stream, "nr_gemm_tn");
}

// The round-3 entry points of the weight-gradient GEMM (X with NR_KP columns): the same general kernel
int nr_tn_gemm_parts(int M, int64_t n_tok) { return nr_gemm_tn_parts(M, NR_KP, n_tok); }
int nr_tn_gemm(const uint16_t* G, int ldg, int M, const uint16_t* X, const uint16_t* zeros, float* out, int64_t n_tok, int P, void* stream) {
  if (!G || !X || !zeros || !out || M <= 0 || M > ldg || (ldg & 7) || n_tok < 0 || P <= 0 || (P & 7)) return fail(NR_ERR_BADARG, "nr_tn_gemm: bad argument");
  return nr_gemm_tn(G, ldg, M, X, NR_KP, NR_KP, 1, zeros, out, NR_KP, n_tok, P, stream);
}

// ---- general-geometry kernels (csrc/k_generic.h): what is not a GEMM on the path for config knobs away from 300 / 15 / 300 / 3 -------------------
int nr_g_dropout(const float* x, float* y, int64_t n_elem, int64_t elem0, float p, uint64_t seed, int site, void* stream) {
  if (!x || !y || n_elem < 0 || (n_elem & 3) || elem0 < 0 || (elem0 & 3) || p < 0.0f || p >= 1.0f || site < 1) return fail(NR_ERR_BADARG, "nr_g_dropout: bad argument");
  if (n_elem == 0) return NR_OK;
  NR_LAUNCH(nr::g_dropout_kernel, grid_for(n_elem / 4, 256, 4096), 256, 0, (hipStream_t)stream, x, y, n_elem / 4, elem0 / 4, make_drop(p, seed), site);
  return check_launch("nr_g_dropout");
}
static bool g_attn_ok(int64_t n_seq, int S, int H, int dk, int64_t ld) {
  return n_seq >= 0 && S >= 1 && S <= nr::G_SMAX && H >= 1 && dk >= 1 && dk <= nr::G_DKMAX && ld >= 3LL * H * dk && n_seq * H < (1LL << 31);
}
int nr_g_attn_fwd(const float* qkv, int64_t ld, float* ctx, const int32_t* key_len, int64_t n_seq, int S, int H, int dk, void* stream) {
  if (!qkv || !ctx || !g_attn_ok(n_seq, S, H, dk, ld)) return fail(NR_ERR_BADARG, "nr_g_attn_fwd: bad argument (S <= 64, d_k <= 32)");
  if (n_seq == 0) return NR_OK;
  nr::GAttnParams p{qkv, ld, nullptr, ctx, key_len, n_seq, S, H, dk, H * dk};
  NR_LAUNCH(nr::g_attn_fwd_kernel, n_seq * H, 64, 2 * S * dk * 4, (hipStream_t)stream, p);
  return check_launch("nr_g_attn_fwd");
}
int nr_g_attn_bwd(const float* qkv, int64_t ld, const float* dctx, float* dqkv, const int32_t* key_len, int64_t n_seq, int S, int H, int dk, void* stream) {
  if (!qkv || !dctx || !dqkv || !g_attn_ok(n_seq, S, H, dk, ld)) return fail(NR_ERR_BADARG, "nr_g_attn_bwd: bad argument (S <= 64, d_k <= 32)");
  if (n_seq == 0) return NR_OK;
  nr::GAttnParams p{qkv, ld, dctx, dqkv, key_len, n_seq, S, H, dk, H * dk};
  NR_LAUNCH(nr::g_attn_bwd_kernel, n_seq * H, 64, (4 * S * dk + 2 * nr::G_SMAX) * 4, (hipStream_t)stream, p);
  return check_launch("nr_g_attn_bwd");
}
int nr_g_additive_fwd(const float* x, int64_t ldx, int D, const float* proj, int64_t ldp, int Q, const float* qv, float* out, int64_t ldo, float* attn_w,
                      int64_t n_seq, int S, int valid, void* stream) {
  if (!x || !proj || !qv || !out || D < 1 || Q < 1 || ldx < D || ldp < Q || ldo < D || n_seq < 0 || S < 1 || S > nr::G_SMAX || valid < 1 || valid > S)
    return fail(NR_ERR_BADARG, "nr_g_additive_fwd: bad argument (S <= 64)");
  if (n_seq == 0) return NR_OK;
  nr::GPoolParams p{};
  p.x = x; p.ldx = ldx; p.D = D; p.proj = proj; p.ldp = ldp; p.Q = Q; p.qv = qv; p.out = out; p.ldo = ldo; p.attn_w = attn_w; p.n_seq = n_seq; p.S = S; p.valid = valid;
  NR_LAUNCH(nr::g_additive_fwd_kernel, n_seq, 256, nr::G_SMAX * 4, (hipStream_t)stream, p);
  return check_launch("nr_g_additive_fwd");
}
int nr_g_additive_bwd(const float* x, int64_t ldx, int D, const float* proj, int64_t ldp, int Q, const float* qv, const float* attn_w, const float* g_out,
                      int64_t ldg, float* dpre, int64_t ldq, float* dq_part, int64_t n_seq, int S, void* stream) {
  if (!x || !proj || !qv || !attn_w || !g_out || !dpre || !dq_part || D < 1 || Q < 1 || ldx < D || ldp < Q || ldg < D || ldq < Q || n_seq < 0 || S < 1 ||
      S > nr::G_SMAX)
    return fail(NR_ERR_BADARG, "nr_g_additive_bwd: bad argument (S <= 64)");
  if (n_seq == 0) return NR_OK;
  nr::GPoolParams p{};
  p.x = x; p.ldx = ldx; p.D = D; p.proj = proj; p.ldp = ldp; p.Q = Q; p.qv = qv; p.g_out = g_out; p.ldg = ldg; p.attn_w = const_cast<float*>(attn_w);
  p.dpre = dpre; p.ldq = ldq; p.dq_part = dq_part; p.n_seq = n_seq; p.S = S; p.valid = S;
  NR_LAUNCH(nr::g_additive_bwd_kernel, n_seq, 256, nr::G_SMAX * 4, (hipStream_t)stream, p);
  return check_launch("nr_g_additive_bwd");
}
int nr_g_rows_axpy(float* y, int64_t ldy, const float* a, const float* g, int64_t ldg, int S, int d, int64_t n_rows, int accumulate, void* stream) {
  if (!y || !a || !g || S < 1 || d < 1 || ldy < d || ldg < d || n_rows < 0) return fail(NR_ERR_BADARG, "nr_g_rows_axpy: bad argument");
  if (n_rows == 0) return NR_OK;
  NR_LAUNCH(nr::g_rows_axpy_kernel, grid_for(n_rows * d, 256, 4096), 256, 0, (hipStream_t)stream, y, ldy, a, g, ldg, S, d, n_rows, accumulate);
  return check_launch("nr_g_rows_axpy");
}
int nr_g_rows_to_seqpad(const float* src, int64_t lds, int d, uint16_t* dst, int dp, int S, int pad, int64_t n_tok, int one, void* stream) {
  if (!src || !dst || d < 1 || dp < d || lds < d || S < 1 || pad < 0 || n_tok < 0 || n_tok % S) return fail(NR_ERR_BADARG, "nr_g_rows_to_seqpad: bad argument");
  if (n_tok == 0) return NR_OK;
  NR_LAUNCH(nr::g_rows_to_seqpad_kernel, grid_for(n_tok * dp, 256, 4096), 256, 0, (hipStream_t)stream, src, lds, d, dst, dp, S, pad, n_tok,
            (uint16_t)(one ? 0x3F80 : 0));
  return check_launch("nr_g_rows_to_seqpad");
}
int nr_g_relu_drop(const float* y, int64_t ldy, float* act, int F, int S, int pad, int64_t n_tok, float p, uint64_t seed, int64_t elem0, void* stream) {
  if (!y || !act || F < 4 || (F & 3) || ldy < F || (ldy & 3) || S < 1 || pad < 0 || n_tok < 0 || n_tok % S || p < 0.0f || p >= 1.0f || elem0 < 0 || (elem0 & 3))
    return fail(NR_ERR_BADARG, "nr_g_relu_drop: bad argument (F and ldy multiples of 4)");
  if (n_tok == 0) return NR_OK;
  NR_LAUNCH(nr::g_relu_drop_kernel, grid_for(n_tok * (F / 4), 256, 4096), 256, 0, (hipStream_t)stream, y, ldy, act, F, S, pad, n_tok, make_drop(p, seed), elem0 / 4);
  return check_launch("nr_g_relu_drop");
}
int nr_g_relu_drop_bwd(const float* dact, const float* act, uint16_t* dy, int F, int fp, int S, int pad, int64_t n_tok, float p, void* stream) {
  if (!dact || !act || !dy || F < 1 || fp < F || S < 1 || pad < 0 || n_tok < 0 || n_tok % S || p < 0.0f || p >= 1.0f) return fail(NR_ERR_BADARG, "nr_g_relu_drop_bwd: bad argument");
  if (n_tok == 0) return NR_OK;
  NR_LAUNCH(nr::g_relu_drop_bwd_kernel, grid_for(n_tok * F, 256, 4096), 256, 0, (hipStream_t)stream, dact, act, dy, F, fp, S, pad, n_tok, 1.0f / (1.0f - p));
  return check_launch("nr_g_relu_drop_bwd");
}
int nr_g_unpad_rows(const float* src, int64_t lds, float* dst, int d, int S, int pad, int64_t n_tok, void* stream) {
  if (!src || !dst || d < 1 || lds < d || S < 1 || pad < 0 || n_tok < 0 || n_tok % S) return fail(NR_ERR_BADARG, "nr_g_unpad_rows: bad argument");
  if (n_tok == 0) return NR_OK;
  NR_LAUNCH(nr::g_unpad_rows_kernel, grid_for(n_tok * d, 256, 4096), 256, 0, (hipStream_t)stream, src, lds, dst, d, S, pad, n_tok);
  return check_launch("nr_g_unpad_rows");
}
int nr_g_rows_split_bf16(const float* src, int64_t ld, int d, uint16_t* dst, int dp, int64_t n, void* stream) {
  if (!src || !dst || d <= 0 || dp <= d || ld < d || n < 0) return fail(NR_ERR_BADARG, "nr_g_rows_split_bf16: bad argument");
  if (n == 0) return NR_OK;
  NR_LAUNCH(nr::g_rows_split_kernel, grid_for(n * dp, 256, 4096), 256, 0, (hipStream_t)stream, src, ld, d, dst, dp, n);
  return check_launch("nr_g_rows_split_bf16");
}
int nr_g_relu(const float* x, const float* gate, float* y, int64_t n, float scale, void* stream) {
  if (!x || !y || n < 0) return fail(NR_ERR_BADARG, "nr_g_relu: bad argument");
  if (n == 0) return NR_OK;
  NR_LAUNCH(nr::g_relu_kernel, grid_for(n, 256, 4096), 256, 0, (hipStream_t)stream, x, gate, y, n, scale);
  return check_launch("nr_g_relu");
}

int nr_pack_conv_dgrad(const float* W, int F, int D, uint16_t* Wd2, void* stream) {
  if (!W || !Wd2 || F <= 0 || F > NR_KP || D <= 0 || D > NR_KP) return fail(NR_ERR_BADARG, "nr_pack_conv_dgrad: bad argument");
  NR_LAUNCH(nr::pack_conv_dgrad_kernel, 300, 256, 0, (hipStream_t)stream, W, F, D, Wd2);
  return check_launch("nr_pack_conv_dgrad");
}

int nr_conv3_dgrad_gemm(const uint16_t* dy_pad, const uint16_t* Wd2, uint16_t* dx, int64_t n_seq, int S, void* stream) {
  if (!dy_pad || !Wd2 || !dx || n_seq < 0 || S < 1) return fail(NR_ERR_BADARG, "nr_conv3_dgrad_gemm: bad argument");
  if ((((uintptr_t)dy_pad | (uintptr_t)Wd2 | (uintptr_t)dx) & 15) != 0) return fail(NR_ERR_BADARG, "nr_conv3_dgrad_gemm: buffers must be 16-byte aligned");
  if (n_seq == 0) return NR_OK;
  if (conv_gemm_persist_ok(n_seq, S)) {
    nr::ConvGemmParams q{};
    q.A = Wd2; q.R = dy_pad; q.C = dx;
    return launch_conv_gemm(q, n_seq, S, stream, "nr_conv3_dgrad_gemm");
  }
  nr::GemmParams p{};
  p.A = Wd2; p.lda = 3 * NR_KP; p.B = dy_pad; p.ldb = NR_KP; p.M = NR_KP; p.K = 3 * NR_KP; p.tapw = 1 << 30;
  const int64_t rows = n_seq * (S + 1) + 1 - 2;                  // virtual rows i = seqpad rows 1 .. rp - 2 (the last seqpad row is a separator)
  if (rows > 0x7FFFFFFF) return fail(NR_ERR_BADARG, "nr_conv3_dgrad_gemm: too many rows");
  p.N = (int)rows; p.Cb = dx; p.S = S;
  using G = nr::GemmGeom<2, 2, 5, 2>;
  p.tiles_m = 1; p.tiles_n = (p.N + G::BN - 1) / G::BN;
  return launch_gemm<2, 2, 5, 2>(p, p.tiles_n, stream, "nr_conv3_dgrad_gemm");
}

int nr_transpose_bf16(const uint16_t* src, int R, int C, int64_t lds, uint16_t* dst, int64_t ldd, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || lds < C || ldd < R) return fail(NR_ERR_BADARG, "nr_transpose_bf16: bad argument");
  NR_LAUNCH2(nr::transpose_bf16_kernel, (C + 31) / 32, (R + 31) / 32, 256, 32 * 33 * 2, (hipStream_t)stream, src, R, C, lds, dst, ldd);
  return check_launch("nr_transpose_bf16");
}

int nr_sum_parts(const float* parts, int P, int64_t n, float* out, int accumulate, void* stream) {
  if (!parts || !out || P <= 0 || n < 0 || (n & 3)) return fail(NR_ERR_BADARG, "nr_sum_parts: bad argument (n must be a multiple of 4)");
  if ((((uintptr_t)parts | (uintptr_t)out) & 15) != 0) return fail(NR_ERR_BADARG, "nr_sum_parts: buffers must be 16-byte aligned");
  if (n == 0) return NR_OK;
  NR_LAUNCH(nr::sum_parts_kernel, grid_for(n / 4, 64, 4096), 256, 4 * 64 * 16, (hipStream_t)stream, parts, P, n / 4, out, accumulate);
  return check_launch("nr_sum_parts");
}

static int attn_fwd_launch(const char* who, const uint16_t* qkv, uint16_t* ctx, const int32_t* key_len, int64_t n_seq, int S, float p_drop,
                           uint64_t seed, const nr::AdditiveParams* pool, void* stream) {
  if (!qkv || !ctx || n_seq < 0) return fail(NR_ERR_BADARG, who, ": bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, who, ": dropout probability out of range");
  if (S != 20) return fail(NR_ERR_UNSUPPORTED, who, ": instantiated for 20-token sequences");
  if (n_seq == 0) return NR_OK;
  nr::AttnFwdParams p;
  p.qkv = qkv; p.ctx = ctx; p.key_len = key_len; p.n_seq = n_seq; p.dc = make_drop(p_drop, seed); p.debug = 0;
  p.pool = nr::AdditiveParams{};
  using G = nr::AttnFwdGeom;
  const int64_t grid = (n_seq + G::TPB - 1) / G::TPB;
  if (allow_smem(nr::attn_fwd_kernel<false, false>, G::SMEM) || allow_smem(nr::attn_fwd_kernel<true, false>, G::SMEM) ||
      allow_smem(nr::attn_fwd_kernel<false, true>, G::SMEM))
    return fail(NR_ERR_LAUNCH, who, ": cannot reserve LDS");
  const char* d = getenv("NR_ATTNF_DEBUG");         // profiling: phase switches (AttnFwdParams::debug), re-read per call
  if (pool != nullptr) {
    p.pool = *pool;
    NR_LAUNCH((nr::attn_fwd_kernel<false, true>), grid, G::WPB * 64, G::SMEM, (hipStream_t)stream, p);
  } else if (d != nullptr && atoi(d) != 0) {
    p.debug = atoi(d);
    NR_LAUNCH((nr::attn_fwd_kernel<true, false>), grid, G::WPB * 64, G::SMEM, (hipStream_t)stream, p);
  } else NR_LAUNCH((nr::attn_fwd_kernel<false, false>), grid, G::WPB * 64, G::SMEM, (hipStream_t)stream, p);
  return check_launch(who);
}

int nr_attn_fwd(const uint16_t* qkv, uint16_t* ctx, const int32_t* key_len, int64_t n_seq, int S, float p_drop, uint64_t seed, void* stream) {
  return attn_fwd_launch("nr_attn_fwd", qkv, ctx, key_len, n_seq, S, p_drop, seed, nullptr, stream);
}

int nr_attn_pool_fwd(const uint16_t* qkv, uint16_t* ctx, const int32_t* key_len, const uint16_t* Wap, const float* bap, const float* qvp,
                     float* out, int64_t out_stride, float* attn_w, int64_t n_seq, int S, int valid, float p_drop, uint64_t seed, void* stream) {
  if (!Wap || !bap || !qvp || !out || valid < 1 || valid > S) return fail(NR_ERR_BADARG, "nr_attn_pool_fwd: bad argument");
  if (out_stride < NR_D || (out_stride & 3)) return fail(NR_ERR_BADARG, "nr_attn_pool_fwd: bad output stride");
  nr::AdditiveParams ap{};
  ap.Wap = Wap; ap.bap = bap; ap.qvp = qvp; ap.out = out; ap.out_stride = out_stride; ap.attn_w = attn_w; ap.valid = valid;
  return attn_fwd_launch("nr_attn_pool_fwd", qkv, ctx, key_len, n_seq, S, p_drop, seed, &ap, stream);
}

int64_t nr_additive_bwd_grid(int64_t n_seq, int S) {
  if (S == 20) return (n_seq + 15) / 16;
  if (S == 50) return pool2_s50(n_seq) ? (n_seq + 3) / 4 : n_seq;
  if (S == 4) return (n_seq + 19) / 20;
  return -1;
}

int nr_pack_additive_t(const float* Wa, int qdim, uint16_t* WaT, void* stream) {
  if (!Wa || !WaT) return fail(NR_ERR_BADARG, "nr_pack_additive_t: null pointer");
  if (qdim <= 0 || qdim > NR_QP) return fail(NR_ERR_UNSUPPORTED, "nr_pack_additive_t: query_vector_dim must be in [1,208]");
  NR_LAUNCH(nr::pack_additive_t_kernel, 64, 256, 0, (hipStream_t)stream, Wa, qdim, WaT);
  return check_launch("nr_pack_additive_t");
}

int nr_additive_bwd_ex(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w,
                       const float* g_out, uint16_t* dpre, float* dq_part, const uint16_t* WaT, uint16_t* dctx, int64_t n_seq, int S,
                       void* stream);

int nr_additive_bwd(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w,
                    const float* g_out, uint16_t* dpre, float* dq_part, int64_t n_seq, int S, void* stream) {
  return nr_additive_bwd_ex(ctx, Wap, bap, qvp, attn_w, g_out, dpre, dq_part, nullptr, nullptr, n_seq, S, stream);
}

int nr_additive_bwd_ex(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w,
                       const float* g_out, uint16_t* dpre, float* dq_part, const uint16_t* WaT, uint16_t* dctx, int64_t n_seq, int S,
                       void* stream) {
  if (!ctx || !Wap || !bap || !qvp || !attn_w || !g_out || !dpre || !dq_part || n_seq < 0 || ((WaT == nullptr) != (dctx == nullptr)))
    return fail(NR_ERR_BADARG, "nr_additive_bwd: bad argument");
  if (n_seq == 0) return NR_OK;
  nr::AdditiveBwdParams p;
  p.ctx = ctx; p.Wap = Wap; p.bap = bap; p.qvp = qvp; p.attn_w = attn_w; p.g_out = g_out; p.dpre = dpre;
  p.dq_part = dq_part; p.WaT = WaT; p.dctx = dctx; p.n_seq = n_seq; p.dy_pad = nullptr; p.act_scale = 1.0f;
  if (S == 20) {                      // (the LDS-tile backward kernels for titles -- 2, 4 and 8 titles per workgroup -- lost to this one in rounds 2 and 3 and are gone)
    if (nr::launch_pool2_bwd(p, (hipStream_t)stream)) return fail(NR_ERR_LAUNCH, "nr_additive_bwd: cannot reserve LDS");
  } else if (S == 50 && pool2_s50(n_seq)) {
    if (nr::launch_pool2_bwd50(p, (hipStream_t)stream)) return fail(NR_ERR_LAUNCH, "nr_additive_bwd: cannot reserve LDS");
  } else if (S == 50) {
    constexpr int NSEQ = 1;
    using G = nr::AddGeom<50, NSEQ>;
    if (allow_smem(nr::additive_bwd_kernel<50, NSEQ>, G::BWD_SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_bwd: cannot reserve LDS");
    NR_LAUNCH((nr::additive_bwd_kernel<50, NSEQ>), (n_seq + NSEQ - 1) / NSEQ, nr::WG, G::BWD_SMEM, (hipStream_t)stream, p);
  } else if (S == 4) {
    constexpr int NSEQ = 20;
    using G = nr::AddGeom<4, NSEQ>;
    if (allow_smem(nr::additive_bwd_kernel<4, NSEQ>, G::BWD_SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_bwd: cannot reserve LDS");
    NR_LAUNCH((nr::additive_bwd_kernel<4, NSEQ>), (n_seq + NSEQ - 1) / NSEQ, nr::WG, G::BWD_SMEM, (hipStream_t)stream, p);
  } else {
    return fail(NR_ERR_UNSUPPORTED, "nr_additive_bwd: sequence length not instantiated (4, 20, 50)");
  }
  return check_launch("nr_additive_bwd");
}

int nr_gather_bf16(const int64_t* ids, const float* table, int64_t num_rows, const float* x_dense, uint16_t* Xb,
                   int64_t n_tokens, float p_drop, uint64_t seed, void* stream) {
  if (!Xb || n_tokens < 0 || (ids == nullptr) == (x_dense == nullptr) || (ids && (!table || num_rows <= 0)))
    return fail(NR_ERR_BADARG, "nr_gather_bf16: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_gather_bf16: dropout probability out of range");
  if (n_tokens == 0) return NR_OK;
  NR_LAUNCH(nr::gather_bf16_kernel, grid_for(n_tokens * (NR_KP / 4), 256, 4096), 256, 0, (hipStream_t)stream, ids, table, num_rows,
            x_dense, Xb, n_tokens, make_drop(p_drop, seed));
  return check_launch("nr_gather_bf16");
}

int nr_embed_scatter_add(const int64_t* ids, const uint16_t* dx, int ldx, float* grad_table, int64_t num_rows, int64_t n_tokens,
                         float p_drop, uint64_t seed, void* stream) {
  if (!ids || !dx || !grad_table || num_rows <= 0 || n_tokens < 0 || ldx < NR_D || (ldx & 3))
    return fail(NR_ERR_BADARG, "nr_embed_scatter_add: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_embed_scatter_add: dropout probability out of range");
  if (n_tokens == 0) return NR_OK;
  NR_LAUNCH(nr::embed_scatter_add_kernel, grid_for(n_tokens * (NR_D / 4), 256, 4096), 256, 0, (hipStream_t)stream, ids, dx, ldx,
            grad_table, num_rows, n_tokens, make_drop(p_drop, seed));
  return check_launch("nr_embed_scatter_add");
}

int nr_embed_scatter_sorted(const int64_t* ids_sorted, const int64_t* perm, const uint16_t* dx, int ldx, float* grad_table,
                            int64_t num_rows, int64_t n_tokens, float p_drop, uint64_t seed, void* stream) {
  if (!ids_sorted || !perm || !dx || !grad_table || num_rows <= 0 || n_tokens < 0 || ldx < NR_D || (ldx & 3) ||
      n_tokens >= (1LL << 31) || num_rows >= (1LL << 31))
    return fail(NR_ERR_BADARG, "nr_embed_scatter_sorted: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_embed_scatter_sorted: dropout probability out of range");
  if (n_tokens == 0) return NR_OK;
  // positions per wave: NR_SCATTER_SPAN (A/B; 64 / 128 / 256 / 512), default 64 -- see the kernel's comment on atomic contention under Zipf ids
  static int span = -1;
  if (span < 0) { const char* e = std::getenv("NR_SCATTER_SPAN"); span = e ? std::atoi(e) : 64; }
  auto go = [&](auto tag) {
    constexpr int SPAN = decltype(tag)::value;
    const int64_t waves = (n_tokens + SPAN - 1) / SPAN;
    NR_LAUNCH((nr::embed_scatter_sorted_kernel<nr::u16, SPAN>), (waves + nr::SC_WAVES - 1) / nr::SC_WAVES, nr::SC_WAVES * 64, nr::SC_SMEM, (hipStream_t)stream,
              ids_sorted, perm, dx, (int64_t)ldx, grad_table, num_rows, n_tokens, make_drop(p_drop, seed), 0);
  };
  if (span == 64) go(nr::IntTag<64>{});
  else if (span == 128) go(nr::IntTag<128>{});
  else if (span == 512) go(nr::IntTag<512>{});
  else if (span == 256) go(nr::IntTag<256>{});
  else go(nr::IntTag<64>{});
  return check_launch("nr_embed_scatter_sorted");
}

int nr_scatter_sorted_f32(const int64_t* ids_sorted, const int64_t* perm, const float* src, int64_t ld, float* dst, int64_t num_rows,
                          int64_t n, int pad_row, void* stream) {
  if (!ids_sorted || !perm || !src || !dst || num_rows <= 0 || n < 0 || ld < NR_D || (ld & 3) || n >= (1LL << 31) ||
      num_rows >= (1LL << 31) || pad_row < -1)
    return fail(NR_ERR_BADARG, "nr_scatter_sorted_f32: bad argument");
  if (n == 0) return NR_OK;
  const int64_t waves = (n + nr::SC_SPAN - 1) / nr::SC_SPAN;
  NR_LAUNCH((nr::embed_scatter_sorted_kernel<float, nr::SC_SPAN>), (waves + nr::SC_WAVES - 1) / nr::SC_WAVES, nr::SC_WAVES * 64, nr::SC_SMEM,
            (hipStream_t)stream, ids_sorted, perm, src, ld, dst, num_rows, n, make_drop(0.0f, 0), pad_row);
  return check_launch("nr_scatter_sorted_f32");
}

int nr_score_dot_bwd(const float* dl, const float* cand, const float* user, float* d_cand, float* d_user, int64_t B, int C, int d,
                     void* stream) {
  if (!dl || !cand || !user || !d_cand || !d_user || B < 0 || C <= 0 || d <= 0 || (d & 3))
    return fail(NR_ERR_BADARG, "nr_score_dot_bwd: bad argument");
  if (B == 0) return NR_OK;
  NR_LAUNCH(nr::score_dot_bwd_kernel, grid_for(B * (d / 4), 256, 2048), 256, 0, (hipStream_t)stream, dl, cand, user, d_cand,
            d_user, B, C, d / 4);
  return check_launch("nr_score_dot_bwd");
}

int nr_score_ce_fwd(const float* cand, const float* user, const int64_t* target, float* logits, float* dl, float* loss_rows, float* loss,
                    int64_t B, int C, int d, void* stream) {
  if (!cand || !user || !dl || !loss_rows || !loss || B <= 0 || C <= 0 || C > 64 || d <= 0 || (d & 3))
    return fail(NR_ERR_BADARG, "nr_score_ce_fwd: bad argument (1 <= C <= 64, d a multiple of 4, B >= 1)");
  if ((((uintptr_t)cand | (uintptr_t)user) & 15) != 0) return fail(NR_ERR_BADARG, "nr_score_ce_fwd: vectors must be 16-byte aligned");
  const float inv_B = 1.0f / (float)B;
  NR_LAUNCH(nr::score_ce_fwd_kernel, (B + 3) / 4, 256, 0, (hipStream_t)stream, cand, user, target, logits, dl, loss_rows, B, C, d / 4, inv_B);
  NR_LAUNCH(nr::score_ce_mean_kernel, 1, 256, 16, (hipStream_t)stream, (const float*)loss_rows, loss, B, inv_B);
  return check_launch("nr_score_ce_fwd");
}

int nr_score_ce_bwd(const float* dl, const float* gscale, const float* cand, const float* user, float* d_cand, int64_t ldc, float* d_user,
                    int64_t ldu, int64_t B, int C, int d, void* stream) {
  if (!dl || !cand || !user || !d_cand || !d_user || B < 0 || C <= 0 || d <= 0 || (d & 3) || ldc < d || ldu < d || (ldc & 3) || (ldu & 3))
    return fail(NR_ERR_BADARG, "nr_score_ce_bwd: bad argument");
  if ((((uintptr_t)cand | (uintptr_t)user | (uintptr_t)d_cand | (uintptr_t)d_user) & 15) != 0)
    return fail(NR_ERR_BADARG, "nr_score_ce_bwd: vectors must be 16-byte aligned");
  if (B == 0) return NR_OK;
  NR_LAUNCH(nr::score_ce_bwd_kernel, grid_for(B * (d / 4), 256, 2048), 256, 0, (hipStream_t)stream, dl, gscale, cand, user, d_cand, ldc,
            d_user, ldu, B, C, d / 4);
  return check_launch("nr_score_ce_bwd");
}

int nr_rows_to_f32(const uint16_t* src, int64_t ld, int d, float* dst, int64_t ldd, int64_t n, void* stream) {
  if (!src || !dst || n < 0 || d <= 0 || (d & 3) || ld < d || ldd < d || (ld & 3) || (ldd & 3))
    return fail(NR_ERR_BADARG, "nr_rows_to_f32: bad argument (d and the row strides must be multiples of 4)");
  if (((uintptr_t)src & 7) != 0 || ((uintptr_t)dst & 15) != 0) return fail(NR_ERR_BADARG, "nr_rows_to_f32: misaligned buffer");
  if (n == 0) return NR_OK;
  NR_LAUNCH(nr::rows_to_f32_kernel, grid_for(n * (d / 4), 256, 8192), 256, 0, (hipStream_t)stream, src, ld, d / 4, dst, ldd, n);
  return check_launch("nr_rows_to_f32");
}

int nr_accum_many(const nr_accum_item* items, int n_items, void* stream) {
  if (n_items < 0 || (n_items > 0 && !items)) return fail(NR_ERR_BADARG, "nr_accum_many: bad argument");
  static_assert(sizeof(nr_accum_item) == sizeof(nr::AccumItem), "nr_accum_item layout");
  for (int base = 0; base < n_items; base += nr::ACCUM_MAX_ITEMS) {
    const int n = n_items - base < nr::ACCUM_MAX_ITEMS ? n_items - base : nr::ACCUM_MAX_ITEMS;
    nr::AccumBatch batch;
    memset(&batch, 0, sizeof(batch));
    int64_t biggest = 0;
    for (int i = 0; i < n; ++i) {
      const nr_accum_item& a = items[base + i];
      if (!a.src || !a.dst || a.rows < 0 || a.cols < 0 || a.src_ld < a.cols || a.dst_ld < a.cols || a.parts < 1 || (a.parts > 1 && a.part_stride < 1))
        return fail(NR_ERR_BADARG, "nr_accum_many: bad item (null pointer, negative extent, a row stride below the row length, parts < 1)");
      batch.it[i] = nr::AccumItem{a.src, a.dst, a.src_ld, a.dst_ld, a.rows, a.cols, a.parts, 0, a.part_stride};
      const int64_t e = (int64_t)a.rows * a.cols * (a.parts > 1 ? 64 : 1);      // (an item with parts covers 16 elements per workgroup pass, not 1024)
      if (e > biggest) biggest = e;
    }
    if (biggest == 0) continue;
    NR_LAUNCH2(nr::accum_many_kernel, grid_for(biggest, 1024, 256), n, 256, nr::ACCUM_PG * 16 * 4, (hipStream_t)stream, batch);
  }
  return check_launch("nr_accum_many");
}

int nr_score_dot(const float* cand, const float* user, float* out, int64_t B, int C, int d, void* stream) {
  if (!cand || !user || !out || B < 0 || C <= 0 || d <= 0 || (d & 3)) return fail(NR_ERR_BADARG, "nr_score_dot: bad argument");
  if (B == 0) return NR_OK;
  int64_t pairs = B * C;
  NR_LAUNCH(nr::score_dot_kernel, (pairs + 3) / 4, 256, 0, (hipStream_t)stream, cand, user, out, B, C, d / 4);
  return check_launch("nr_score_dot");
}

int nr_score_csr(const float* news, const float* users, const int32_t* cand_idx, const int64_t* cand_ptr,
                 const int32_t* user_idx, float* out, int64_t n_impr, int64_t nnz, int d, void* stream) {
  if (!news || !users || !cand_idx || !cand_ptr || !user_idx || !out || n_impr < 0 || nnz < 0 || d <= 0 || (d & 3))
    return fail(NR_ERR_BADARG, "nr_score_csr: bad argument");
  if (nnz == 0 || n_impr == 0) return NR_OK;
  NR_LAUNCH(nr::score_csr_kernel, (nnz + 3) / 4, 256, 0, (hipStream_t)stream, news, users, cand_idx, cand_ptr, user_idx, out,
            n_impr, nnz, d / 4);
  return check_launch("nr_score_csr");
}

// ---- convolutional text encoder (NAML / LSTUR) ----------------------------------------------------------------------
int nr_pack_conv(const float* W, const float* b, int F, int D, uint16_t* Wc, uint16_t* Wd, float* bc, void* stream) {
  if (!W || !b || !Wc || !bc) return fail(NR_ERR_BADARG, "nr_pack_conv: null pointer");
  if (F <= 0 || F > NR_D || D <= 0 || D > NR_D) return fail(NR_ERR_UNSUPPORTED, "nr_pack_conv: num_filters and word_embedding_dim must be <= 300");
  NR_LAUNCH(nr::pack_conv_kernel, 256, 256, 0, (hipStream_t)stream, W, b, F, D, Wc, Wd, bc);
  return check_launch("nr_pack_conv");
}

static int launch_conv(nr::ConvParams& p, int S, void* stream, const char* what) {
  { static int dbg = -1; if (dbg < 0) { const char* d = getenv("NR_CONV_DEBUG"); dbg = d ? atoi(d) : 0; } p.debug = dbg; }
  // 8 waves on 8 titles / 4 abstracts: one workgroup per CU, the filter bank is re-read from L2 half as often as with 4 waves on 4 / 2
  // (~10 % at B = 512, profiles/r02_ab_switches.txt; the 4-wave instantiations are gone)
  // NR_CONV_HALF_TILE=1 (A/B, read once): half the sequences per workgroup -> two workgroups of 8 waves per CU (70 / 57 KB of LDS each), one's gather
  // phase beside the other's GEMM phase.  Measured in round 6 (profiles/r06_ab_conv_half_tile.txt): 1,483 -> 1,800 us stand-alone for the abstracts,
  // 670 -> 818 us for the titles -- every workgroup streams the 614 KB filter bank from L2, and twice as many workgroups stream it twice as often
  // (~0.3 ms of the kernel is that stream).  Not the default.
  static const int half = [] { const char* e = getenv("NR_CONV_HALF_TILE"); return e ? atoi(e) : 0; }();
  int rc;
  if (S == 20) {
    rc = half ? launch_conv_t<20, 4, 8>(p, stream) : launch_conv_t<20, 8, 8>(p, stream);
    if (rc) return rc;
  } else if (S == 50) {
    rc = half ? launch_conv_t<50, 2, 8>(p, stream) : launch_conv_t<50, 4, 8>(p, stream);
    if (rc) return rc;
  } else {
    return fail(NR_ERR_UNSUPPORTED, "conv3: sequence length not instantiated (20, 50)");
  }
  return check_launch(what);
}

int nr_conv3_fwd_v(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wc, const float* bc, uint16_t* act,
                   uint16_t* x_save, int64_t n_seq, int S, int valid, float p_drop, uint64_t seed, int64_t tok_offset, void* stream);

int nr_conv3_fwd(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wc, const float* bc, uint16_t* act,
                 uint16_t* x_save, int64_t n_seq, int S, float p_drop, uint64_t seed, int64_t tok_offset, void* stream) {
  return nr_conv3_fwd_v(ids, table, num_rows, Wc, bc, act, x_save, n_seq, S, S, p_drop, seed, tok_offset, stream);
}

int nr_conv3_fwd_v(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wc, const float* bc, uint16_t* act,
                   uint16_t* x_save, int64_t n_seq, int S, int valid, float p_drop, uint64_t seed, int64_t tok_offset, void* stream) {
  if (!ids || !table || num_rows <= 0 || !Wc || !bc || !act || n_seq < 0 || tok_offset < 0 || valid < 1 || valid > S)
    return fail(NR_ERR_BADARG, "nr_conv3_fwd: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_conv3_fwd: dropout probability out of range");
  if (n_seq == 0) return NR_OK;
  nr::ConvParams p;
  p.ids = ids; p.table = table; p.num_rows = num_rows; p.x_pad = nullptr; p.Wc = Wc; p.bc = bc; p.out = act; p.x_save = x_save;
  p.relu_drop = 1; p.n_seq = n_seq; p.tok_offset = tok_offset; p.valid = valid; p.dc = make_drop(p_drop, seed);
  return launch_conv(p, S, stream, "nr_conv3_fwd");
}

int nr_pack_conv_fwd2(const float* W, int F, int D, uint16_t* Wf2, void* stream) {
  if (!W || !Wf2 || F <= 0 || F > NR_D || D <= 0 || D > NR_D) return fail(NR_ERR_BADARG, "nr_pack_conv_fwd2: bad argument");
  NR_LAUNCH(nr::pack_conv_fwd2_kernel, 300, 256, 0, (hipStream_t)stream, W, F, D, Wf2);
  return check_launch("nr_pack_conv_fwd2");
}

// The training forward of the text encoders as gather pass + persistent ring GEMM (csrc/k_convgemm.h, EPI): same outputs as nr_conv3_fwd_v with
// x_save (which is the GEMM's token-row operand here, hence required).
int nr_conv3_fwd_gemm(const int64_t* ids, const float* table, int64_t num_rows, const uint16_t* Wf2, const float* bc, uint16_t* act,
                      uint16_t* x_save, int64_t n_seq, int S, int valid, float p_drop, uint64_t seed, int64_t tok_offset, void* stream) {
  if (!ids || !table || num_rows <= 0 || !Wf2 || !bc || !act || !x_save || n_seq < 0 || tok_offset < 0 || S < 1 || valid < 1 || valid > S)
    return fail(NR_ERR_BADARG, "nr_conv3_fwd_gemm: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_conv3_fwd_gemm: dropout probability out of range");
  if ((((uintptr_t)Wf2 | (uintptr_t)act | (uintptr_t)x_save | (uintptr_t)table) & 15) != 0) return fail(NR_ERR_BADARG, "nr_conv3_fwd_gemm: buffers must be 16-byte aligned");
  if (n_seq == 0) return NR_OK;
  if (n_seq * S * (int64_t)(NR_KP * 2) >= (1LL << 31) || n_seq * (S + 1) >= (1LL << 31))
    return fail(NR_ERR_UNSUPPORTED, "nr_conv3_fwd_gemm: more than 2 GiB of output rows (use nr_conv3_fwd_v)");
  using G = nr::ConvGemmGeom;
  nr::ConvGatherParams g;
  g.ids = ids; g.table = table; g.num_rows = num_rows; g.x_save = x_save; g.n_seq = n_seq; g.S = S; g.valid = valid; g.tok_offset = tok_offset;
  g.dc = make_drop(p_drop, seed);
  NR_LAUNCH(nr::conv_gather_kernel, grid_for((n_seq * (S + 1) + 1) * (NR_KP / 4), 256, 16384), 256, 0, (hipStream_t)stream, g);
  nr::ConvGemmParams q{};
  q.A = Wf2; q.R = x_save; q.C = act; q.bias = bc; q.tok_offset = tok_offset; q.dc = g.dc;
  q.n_rows = n_seq * (S + 1) - 1; q.n_tok = n_seq * S;
  q.n_tiles = (int)((q.n_rows + G::BN - 1) / G::BN);
  q.S1 = (uint32_t)(S + 1); q.s1_magic = (uint32_t)((1ULL << 32) / (uint32_t)(S + 1)) + 1u;
  const int cus = nr::device_cus();
  if (allow_smem(nr::conv_gemm_kernel<false, false, true, true>, G::SMEM_EPI)) return fail(NR_ERR_LAUNCH, "nr_conv3_fwd_gemm: cannot reserve LDS");
  NR_LAUNCH((nr::conv_gemm_kernel<false, false, true, true>), q.n_tiles < cus ? q.n_tiles : cus, 512, G::SMEM_EPI, (hipStream_t)stream, q);
  return check_launch("nr_conv3_fwd_gemm");
}

int nr_conv3_dgrad(const uint16_t* dy_pad, const uint16_t* Wd, uint16_t* dx, int64_t n_seq, int S, void* stream) {
  if (!dy_pad || !Wd || !dx || n_seq < 0) return fail(NR_ERR_BADARG, "nr_conv3_dgrad: bad argument");
  if (n_seq == 0) return NR_OK;
  nr::ConvParams p;
  p.ids = nullptr; p.table = nullptr; p.num_rows = 0; p.x_pad = dy_pad; p.Wc = Wd; p.bc = nullptr; p.out = dx; p.x_save = nullptr;
  p.relu_drop = 0; p.n_seq = n_seq; p.tok_offset = 0; p.valid = 0; p.dc = make_drop(0.0f, 0);
  return launch_conv(p, S, stream, "nr_conv3_dgrad");
}

int nr_conv_act_bwd(const uint16_t* act, const uint16_t* dact_gemm, int ldc, const float* attn_w, const float* g_out, int64_t g_stride,
                    uint16_t* dy_pad, int64_t n_seq, int S, float p_drop, void* stream) {
  if (!act || !dact_gemm || !attn_w || !g_out || !dy_pad || n_seq < 0 || S <= 0 || ldc < NR_D || (ldc & 3) || g_stride < NR_D || (g_stride & 3))
    return fail(NR_ERR_BADARG, "nr_conv_act_bwd: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_conv_act_bwd: dropout probability out of range");
  if (n_seq == 0) return NR_OK;
  NR_LAUNCH(nr::conv_act_bwd_kernel, grid_for(n_seq * S * (NR_D / 4), 256, 8192), 256, 0, (hipStream_t)stream, act, dact_gemm, ldc, attn_w,
            g_out, g_stride, dy_pad, n_seq, S, 1.0f / (1.0f - p_drop));
  return check_launch("nr_conv_act_bwd");
}

int nr_additive_bwd_act(const uint16_t* act, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w, const float* g_out,
                        uint16_t* dpre, float* dq_part, const uint16_t* WaT, uint16_t* dctx_scratch, uint16_t* dy_pad, float p_drop,
                        int64_t n_seq, int S, void* stream) {
  if (!act || !Wap || !bap || !qvp || !attn_w || !g_out || !dpre || !dq_part || !WaT || !dctx_scratch || !dy_pad || n_seq < 0)
    return fail(NR_ERR_BADARG, "nr_additive_bwd_act: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_additive_bwd_act: dropout probability out of range");
  if (n_seq == 0) return NR_OK;
  const bool reg = S == 20 || (S == 50 && pool2_s50(n_seq));      // the register-resident kernels fuse the activation gradient
  if (reg) {
    nr::AdditiveBwdParams p;
    p.ctx = act; p.Wap = Wap; p.bap = bap; p.qvp = qvp; p.attn_w = attn_w; p.g_out = g_out; p.dpre = dpre; p.dq_part = dq_part; p.WaT = WaT;
    p.dctx = nullptr; p.n_seq = n_seq; p.dy_pad = dy_pad; p.act_scale = 1.0f / (1.0f - p_drop);
    if (S == 20 ? nr::launch_pool2_bwd(p, (hipStream_t)stream) : nr::launch_pool2_bwd50(p, (hipStream_t)stream))
      return fail(NR_ERR_LAUNCH, "nr_additive_bwd_act: cannot reserve LDS");
    return check_launch("nr_additive_bwd_act");
  }
  const int rc = nr_additive_bwd_ex(act, Wap, bap, qvp, attn_w, g_out, dpre, dq_part, WaT, dctx_scratch, n_seq, S, stream);
  if (rc) return rc;
  return nr_conv_act_bwd(act, dctx_scratch, NR_KP, attn_w, g_out, NR_D, dy_pad, n_seq, S, p_drop, stream);
}

static unsigned long long* g_pool3_stamps = nullptr;
int nr_debug_pool3_stamps(uint64_t* buf) { g_pool3_stamps = (unsigned long long*)buf; return NR_OK; }

int64_t nr_additive_bwd_flat_grid(int64_t n_tok) {
  if (n_tok <= 0) return 0;
  const int64_t wgs = ((n_tok + nr::Pool3Geom::ROWS - 1) / nr::Pool3Geom::ROWS + nr::Pool3Geom::NWAVE - 1) / nr::Pool3Geom::NWAVE;
  const int64_t cus = nr::device_cus();
  return wgs < cus ? wgs : cus;
}

int nr_additive_bwd_flat(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w, const float* g_out,
                         const float* y, int64_t y_stride, float* tot, uint16_t* dpre, float* dq_part, uint16_t* dctx, uint16_t* dy_pad,
                         float p_drop, int64_t n_seq, int S, int qdim, void* stream) {
  return nr_additive_bwd_flat_gs(ctx, Wap, bap, qvp, attn_w, g_out, NR_D, y, y_stride, tot, dpre, dq_part, dctx, dy_pad, p_drop, n_seq, S, qdim, stream);
}

int nr_additive_bwd_flat_gs(const uint16_t* ctx, const uint16_t* Wap, const float* bap, const float* qvp, const float* attn_w, const float* g_out,
                            int64_t g_stride, const float* y, int64_t y_stride, float* tot, uint16_t* dpre, float* dq_part, uint16_t* dctx,
                            uint16_t* dy_pad, float p_drop, int64_t n_seq, int S, int qdim, void* stream) {
  if (!ctx || !Wap || !bap || !qvp || !attn_w || !g_out || !y || !tot || !dpre || !dq_part || n_seq < 0 || (dctx && dy_pad) ||
      y_stride < NR_D || (y_stride & 3) || g_stride < NR_D || (g_stride & 3) || ((uintptr_t)g_out & 15) != 0)
    return fail(NR_ERR_BADARG, "nr_additive_bwd_flat: bad argument");
  if (p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_additive_bwd_flat: dropout probability out of range");
  if (qdim < 1 || qdim > NR_QP) return fail(NR_ERR_BADARG, "nr_additive_bwd_flat: query_vector_dim out of range");
  if (qdim > nr::Pool3Geom::WROWS)
    return fail(NR_ERR_UNSUPPORTED, "nr_additive_bwd_flat: the flat kernel keeps 200 rows of Wa in LDS (query_vector_dim <= 200); use nr_additive_bwd_ex / _act");
  // 48 consecutive tokens belong to 1 + ceil(47 / S) sequences: the kernel has 8 slots for them (16 below S = 7), 4 in the activation-gradient form
  if (S < 4 || (dy_pad && S < 16) || n_seq * (int64_t)S >= (1LL << 31) || n_seq * (g_stride * 4) >= (1LL << 31))
    return fail(NR_ERR_UNSUPPORTED, "nr_additive_bwd_flat: sequence length must be >= 4 (>= 16 with dy_pad), n_seq * S < 2^31 and n_seq * g_stride < 2^29");
  if (n_seq == 0) return NR_OK;
  NR_LAUNCH(nr::rowdot_kernel, grid_for(n_seq, 4, 4096), 256, 0, (hipStream_t)stream, g_out, g_stride, y, y_stride, n_seq, NR_D, tot);
  nr::Pool3Params p;
  p.ctx = ctx; p.Wap = Wap; p.bap = bap; p.qvp = qvp; p.attn_w = attn_w; p.g_out = g_out; p.g_row_bytes = (uint32_t)(g_stride * 4);
  p.tot = tot; p.dpre = dpre; p.dq_part = dq_part;
  p.dctx = dctx; p.dy_pad = dy_pad; p.act_scale = 1.0f / (1.0f - p_drop); p.n_seq = n_seq; p.n_tok = n_seq * S; p.S = (uint32_t)S;
  p.s_magic = (uint32_t)((1ULL << 32) / (uint32_t)S) + 1u;
  const int64_t grid = nr_additive_bwd_flat_grid(p.n_tok);
  const char* d = getenv("NR_POOL_DEBUG");        // profiling: phase switches of the DBG instantiations (re-read per call; tools/pool_phases.sh)
  p.dbg = d ? atoi(d) : 0;
  p.stamps = g_pool3_stamps;
#define NR_POOL3_LAUNCH(ACT_, DBG_)                                                                                                       \
  do {                                                                                                                                    \
    if (allow_smem(nr::pool3_bwd_kernel<ACT_, DBG_>, nr::Pool3Geom::SMEM)) return fail(NR_ERR_LAUNCH, "nr_additive_bwd_flat: cannot reserve LDS"); \
    NR_LAUNCH((nr::pool3_bwd_kernel<ACT_, DBG_>), grid, nr::Pool3Geom::THREADS, nr::Pool3Geom::SMEM, (hipStream_t)stream, p);      \
  } while (0)
  if (dy_pad) { if (p.dbg) NR_POOL3_LAUNCH(true, true); else NR_POOL3_LAUNCH(true, false); }
  else { if (p.dbg) NR_POOL3_LAUNCH(false, true); else NR_POOL3_LAUNCH(false, false); }
#undef NR_POOL3_LAUNCH
  return check_launch("nr_additive_bwd_flat");
}

int nr_additive_dx(const uint16_t* dgemm, int ldc, const float* attn_w, const float* g_out, float* dx, int64_t n_seq, int S,
                   int view_major, void* stream) {
  if (!dgemm || !attn_w || !g_out || !dx || n_seq < 0 || S <= 0 || ldc < NR_D || (ldc & 3)) return fail(NR_ERR_BADARG, "nr_additive_dx: bad argument");
  if (n_seq == 0) return NR_OK;
  NR_LAUNCH(nr::additive_dx_kernel, grid_for(n_seq * S * (NR_D / 4), 256, 8192), 256, 0, (hipStream_t)stream, dgemm, ldc, attn_w, g_out, dx,
            n_seq, S, view_major);
  return check_launch("nr_additive_dx");
}

int nr_element_table_fwd(const float* emb, int ncat, int dcat, const float* W0, const float* b0, const float* W1, const float* b1, float* E,
                         void* stream) {
  if (!emb || !W0 || !b0 || !W1 || !b1 || !E || ncat <= 0 || dcat <= 0) return fail(NR_ERR_BADARG, "nr_element_table_fwd: bad argument");
  NR_LAUNCH(nr::element_table_fwd_kernel, grid_for(2LL * ncat * NR_D * nr::ET_LANES, 256, 8192), 256, 0, (hipStream_t)stream, emb, ncat, dcat, W0, b0, W1, b1, E, NR_D);
  return check_launch("nr_element_table_fwd");
}

int nr_element_table_bwd(const float* emb, int ncat, int dcat, const float* W0, const float* W1, const float* E, const float* dE, float* dW,
                         float* db, float* demb, void* stream) {
  if (!emb || !W0 || !W1 || !E || !dE || !dW || !db || !demb || ncat <= 0 || dcat <= 0) return fail(NR_ERR_BADARG, "nr_element_table_bwd: bad argument");
  NR_LAUNCH(nr::element_table_bwd_kernel, grid_for((2LL * NR_D * dcat + 2 * NR_D + (int64_t)ncat * dcat) * nr::ET_LANES, 256, 8192), 256, 0, (hipStream_t)stream,
            emb, ncat, dcat, W0, W1, E, dE, NR_D, dW, db, demb);
  return check_launch("nr_element_table_bwd");
}

int nr_views_fill(const int64_t* cat, const int64_t* sub, const float* E, int ncat, uint16_t* views, int64_t T, void* stream) {
  if (!cat || !sub || !E || !views || ncat <= 0 || T < 0) return fail(NR_ERR_BADARG, "nr_views_fill: bad argument");
  if (T == 0) return NR_OK;
  NR_LAUNCH(nr::views_fill_kernel, grid_for(T * 2 * (NR_KP / 4), 256, 4096), 256, 0, (hipStream_t)stream, cat, sub, E, ncat, views, T);
  return check_launch("nr_views_fill");
}

int nr_rows_scatter_add(const int64_t* ids, const float* src, int64_t ld, const float* row_scale, float* dst, int64_t num_rows, int d,
                        int64_t n, int pad_row, void* stream) {
  if (!ids || !src || !dst || num_rows <= 0 || d <= 0 || ld < d || n < 0) return fail(NR_ERR_BADARG, "nr_rows_scatter_add: bad argument");
  if (n == 0) return NR_OK;
  NR_LAUNCH(nr::rows_scatter_add_kernel, grid_for(n * d, 256, 4096), 256, 0, (hipStream_t)stream, ids, src, ld, row_scale, dst, num_rows, d, n, pad_row);
  return check_launch("nr_rows_scatter_add");
}

int nr_gather_rows_strided(const int64_t* ids, const float* table, int64_t num_rows, int d, const float* row_scale, float* out, int64_t ldo,
                           int64_t n, void* stream) {
  if (!ids || !table || !out || num_rows <= 0 || d <= 0 || ldo < d || n < 0) return fail(NR_ERR_BADARG, "nr_gather_rows_strided: bad argument");
  if (n == 0) return NR_OK;
  if ((d & 3) || (ldo & 3) || ((uintptr_t)out & 15) || ((uintptr_t)table & 15)) {
    NR_LAUNCH(nr::gather_rows_strided_scalar_kernel, grid_for(n * d, 256, 4096), 256, 0, (hipStream_t)stream, ids, table, num_rows, d, row_scale, out,
              ldo, n);
    return check_launch("nr_gather_rows_strided");
  }
  NR_LAUNCH(nr::gather_rows_strided_kernel, grid_for(n * (d / 4), 256, 4096), 256, 0, (hipStream_t)stream, ids, table, num_rows, d, row_scale, out, ldo, n);
  return check_launch("nr_gather_rows_strided");
}

// ---- LSTUR GRU user encoder --------------------------------------------------------------------------------------------
static inline int ceil_to(int v, int m) { return (v + m - 1) / m * m; }

int nr_gru_dims(int Hd, int* Hg, int* Hp, int* Kp) {
  if (Hd <= 0 || !Hg || !Hp || !Kp) return fail(NR_ERR_BADARG, "nr_gru_dims: bad argument");
  *Hg = ceil_to(Hd, 16); *Hp = ceil_to(Hd + 1, 32); *Kp = ceil_to(3 * *Hg, 32);
  return NR_OK;
}

int nr_pack_gru(const float* W, int Hd, int K, int Kpad, uint16_t* dst, uint16_t* dstT, int tiled, void* stream) {
  if (!W || !dst || Hd <= 0 || K <= 0 || Kpad < K || (Kpad & 31)) return fail(NR_ERR_BADARG, "nr_pack_gru: bad argument");
  const int Hg = ceil_to(Hd, 16), Kp = ceil_to(3 * Hg, 32);
  if (tiled && !(Hg & 15))          // (3 Hg rows and Kpad / Kp columns are whole 16 x 32 blocks: the image is a sequence of 16-byte pieces)
    NR_LAUNCH(nr::pack_gru_tiled_kernel, 2048, 256, 0, (hipStream_t)stream, W, Hd, K, Hg, Kpad, dst, dstT, Kpad, Kp);
  else
    NR_LAUNCH(nr::pack_gru_kernel, 1024, 256, 0, (hipStream_t)stream, W, Hd, K, Hg, Kpad, dst, dstT, Kpad, Kp, tiled);
  return check_launch("nr_pack_gru");
}

int nr_rows_to_bf16(const float* src, int64_t ld, int d, uint16_t* dst, int dp, int64_t n, void* stream) {
  if (!src || !dst || d <= 0 || dp < d || ld < d || n < 0) return fail(NR_ERR_BADARG, "nr_rows_to_bf16: bad argument");
  if (n == 0) return NR_OK;
  if (!(d & 3) && !(dp & 3) && !(ld & 3) && (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 7) == 0))
    NR_LAUNCH(nr::rows_to_bf16_v4_kernel, grid_for(n * (dp / 4), 256, 8192), 256, 0, (hipStream_t)stream, src, ld, d, dst, dp, n);
  else
    NR_LAUNCH(nr::rows_to_bf16_kernel, grid_for(n * dp, 256, 8192), 256, 0, (hipStream_t)stream, src, ld, d, dst, dp, n);
  return check_launch("nr_rows_to_bf16");
}

int nr_tile_rows_bf16(const uint16_t* src, int n, int K, uint16_t* dst, void* stream) {
  if (!src || !dst || n < 0 || K <= 0 || (K & 31)) return fail(NR_ERR_BADARG, "nr_tile_rows_bf16: bad argument");
  if (n == 0) return NR_OK;
  NR_LAUNCH(nr::tile_rows_kernel, grid_for((int64_t)ceil_to(n, 16) * K, 256, 8192), 256, 0, (hipStream_t)stream, src, n, K, dst);
  return check_launch("nr_tile_rows_bf16");
}

// GRU tuning knobs, read once.  NR_GRU_NB: sample tiles per wave in the forward step (default 2 from 256 samples up); NR_GRU_LDS=0:
// register-only step kernels instead of the W_hh-tile-in-LDS ones (Hd = 900 / 450).
static int gru_nb_knob() { static int v = -1; if (v < 0) { const char* e = getenv("NR_GRU_NB"); v = e ? atoi(e) : 0; } return v; }
static int gru_lds_knob() { static int v = -1; if (v < 0) { const char* e = getenv("NR_GRU_LDS"); v = e ? atoi(e) : 1; } return v; }

static int gru_fwd_step_launch(const float* gi, const int32_t* gi_row, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len,
                               const uint16_t* h_in_t, uint16_t* h_out_b, uint16_t* h_out_t, const float* h_in_f, float* h_out_f, uint16_t* gates, int B,
                               int N, int Hd, int t, void* stream);

int nr_gru_fwd_step(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, const uint16_t* h_in_t,
                    uint16_t* h_out_b, uint16_t* h_out_t, const float* h_in_f, float* h_out_f, uint16_t* gates, int B, int N, int Hd, int t,
                    void* stream) {
  return gru_fwd_step_launch(gi, nullptr, Whh, b_ih, b_hh, len, h_in_t, h_out_b, h_out_t, h_in_f, h_out_f, gates, B, N, Hd, t, stream);
}

static int gru_fwd_step_launch(const float* gi, const int32_t* gi_row, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len,
                               const uint16_t* h_in_t, uint16_t* h_out_b, uint16_t* h_out_t, const float* h_in_f, float* h_out_f, uint16_t* gates, int B,
                               int N, int Hd, int t, void* stream) {
  if (!gi || !Whh || !b_ih || !b_hh || !len || !h_in_t || !h_out_t || !h_in_f || !h_out_f || B < 0 || N <= 0 || Hd <= 0 || t < 0 || t >= N)
    return fail(NR_ERR_BADARG, "nr_gru_fwd_step: bad argument");
  if (B == 0) return NR_OK;
  nr::GruFwdParams p;
  p.gi = gi; p.gi_row = gi_row; p.Whh = Whh; p.b_ih = b_ih; p.b_hh = b_hh; p.len = len; p.h_in_t = h_in_t; p.h_out_b = h_out_b; p.h_out_t = h_out_t;
  p.h_in_f = h_in_f; p.h_out_f = h_out_f; p.gates = gates; p.B = B; p.N = N; p.Hd = Hd; p.Hg = ceil_to(Hd, 16); p.Hp = ceil_to(Hd + 1, 32); p.t = t;
  const int nb = gru_nb_knob() > 0 ? gru_nb_knob() : (B >= 256 ? 2 : 1);
  const int tiles = p.Hg / 16;
  const int ldsv = gru_lds_knob();      // W_hh-tile-in-LDS variant of the two-tile kernel (Hd = 900 / 450): the default
  if (ldsv == 1 && nb == 2 && (p.Hp == 29 * 32 || p.Hp == 15 * 32)) {
    const int grid = nr::gru_grid(tiles, (B + 127) / 128), smem = 3 * p.Hp * 16 * 2;
    if (p.Hp == 29 * 32) {
      if (allow_smem(nr::gru_fwd_step_kernel<29, 2, true>, smem)) return fail(NR_ERR_LAUNCH, "nr_gru_fwd_step: cannot reserve LDS");
      NR_LAUNCH2((nr::gru_fwd_step_kernel<29, 2, true>), grid, 1, nr::WG, smem, (hipStream_t)stream, p);
    } else {
      if (allow_smem(nr::gru_fwd_step_kernel<15, 2, true>, smem)) return fail(NR_ERR_LAUNCH, "nr_gru_fwd_step: cannot reserve LDS");
      NR_LAUNCH2((nr::gru_fwd_step_kernel<15, 2, true>), grid, 1, nr::WG, smem, (hipStream_t)stream, p);
    }
  } else if (nb == 2) {
    const int grid = nr::gru_grid(tiles, (B + 127) / 128);
    if (p.Hp == 29 * 32) NR_LAUNCH2((nr::gru_fwd_step_kernel<29, 2>), grid, 1, nr::WG, 0, (hipStream_t)stream, p);          // Hd = 900
    else if (p.Hp == 15 * 32) NR_LAUNCH2((nr::gru_fwd_step_kernel<15, 2>), grid, 1, nr::WG, 0, (hipStream_t)stream, p);     // Hd = 450
    else NR_LAUNCH2((nr::gru_fwd_step_kernel<0, 2>), grid, 1, nr::WG, 0, (hipStream_t)stream, p);
  } else {
    const int grid = nr::gru_grid(tiles, (B + 63) / 64);
    if (p.Hp == 29 * 32) NR_LAUNCH2((nr::gru_fwd_step_kernel<29, 1>), grid, 1, nr::WG, 0, (hipStream_t)stream, p);
    else if (p.Hp == 15 * 32) NR_LAUNCH2((nr::gru_fwd_step_kernel<15, 1>), grid, 1, nr::WG, 0, (hipStream_t)stream, p);
    else NR_LAUNCH2((nr::gru_fwd_step_kernel<0, 1>), grid, 1, nr::WG, 0, (hipStream_t)stream, p);
  }
  return check_launch("nr_gru_fwd_step");
}

int nr_gru_bwd_step(const float* g_last, const uint16_t* dgh_next, const float* carry_next, const uint16_t* WhhT, const uint16_t* gates,
                    const uint16_t* h_prev_b, const int32_t* len, uint16_t* dgi, uint16_t* dgh, uint16_t* dgh_t, float* carry, int B, int N,
                    int Hd, int t, int first, void* stream) {
  if (!WhhT || !len || !carry || B < 0 || N <= 0 || Hd <= 0 || t < -1 || t >= N) return fail(NR_ERR_BADARG, "nr_gru_bwd_step: bad argument");
  if (first ? !g_last : (!dgh_next || !carry_next)) return fail(NR_ERR_BADARG, "nr_gru_bwd_step: missing upstream gradient");
  if (t >= 0 && (!gates || !h_prev_b || !dgi || !dgh || !dgh_t)) return fail(NR_ERR_BADARG, "nr_gru_bwd_step: missing step buffers");
  if (B == 0) return NR_OK;
  nr::GruBwdParams p;
  p.g_last = g_last; p.dgh_next = dgh_next; p.carry_next = carry_next; p.WhhT = WhhT; p.gates = t >= 0 ? gates : nullptr; p.h_prev_b = h_prev_b;
  p.len = len; p.dgi = dgi; p.dgh = dgh; p.dgh_t = dgh_t; p.carry = carry; p.B = B; p.N = N; p.Hd = Hd; p.Hg = ceil_to(Hd, 16); p.Hp = ceil_to(Hd + 1, 32);
  p.Kp = ceil_to(3 * p.Hg, 32); p.t = t; p.first = first;
  const int ldsv = gru_lds_knob();      // W_hh^T-tile-in-LDS variant (Hd = 900 / 450): default, NR_GRU_LDS=0 selects the register-only kernel
  if (ldsv == 1 && (p.Kp == 86 * 32 || p.Kp == 44 * 32)) {
    const int grid = nr::gru_grid(p.Hg / 16, (B + 127) / 128), smem = p.Kp * 16 * 2;
    if (p.Kp == 86 * 32) {
      if (allow_smem(nr::gru_bwd_step_lds_kernel<86>, smem)) return fail(NR_ERR_LAUNCH, "nr_gru_bwd_step: cannot reserve LDS");
      NR_LAUNCH2(nr::gru_bwd_step_lds_kernel<86>, grid, 1, nr::WG, smem, (hipStream_t)stream, p);
    } else {
      if (allow_smem(nr::gru_bwd_step_lds_kernel<44>, smem)) return fail(NR_ERR_LAUNCH, "nr_gru_bwd_step: cannot reserve LDS");
      NR_LAUNCH2(nr::gru_bwd_step_lds_kernel<44>, grid, 1, nr::WG, smem, (hipStream_t)stream, p);
    }
    return check_launch("nr_gru_bwd_step");
  }
  if (p.Kp == 86 * 32) NR_LAUNCH2(nr::gru_bwd_step_kernel<86>, nr::gru_grid(p.Hg / 16, (B + 63) / 64), 1, nr::WG, 0, (hipStream_t)stream, p);          // Hd = 900
  else if (p.Kp == 44 * 32) NR_LAUNCH2(nr::gru_bwd_step_kernel<44>, nr::gru_grid(p.Hg / 16, (B + 63) / 64), 1, nr::WG, 0, (hipStream_t)stream, p);     // Hd = 450
  else NR_LAUNCH2(nr::gru_bwd_step_kernel<0>, nr::gru_grid(p.Hg / 16, (B + 63) / 64), 1, nr::WG, 0, (hipStream_t)stream, p);
  return check_launch("nr_gru_bwd_step");
}

#ifndef NR_EMU
int nr_debug_xcd_probe(uint32_t* sync_words, uint32_t* rec, uint32_t* out, int phases, void* stream) {
  if (!sync_words || !rec || !out || phases < 1 || phases > 1000) return fail(NR_ERR_BADARG, "nr_debug_xcd_probe: bad argument");
  if (hipMemsetAsync(sync_words, 0, 32 * sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) return fail(NR_ERR_LAUNCH, "nr_debug_xcd_probe: memset");
  nr::XcdSync s{sync_words, nullptr, 0};
  NR_LAUNCH(nr::xcd_probe_kernel, nr::NR_XCDS * nr::NR_XCD_TEAM, 512, 0, (hipStream_t)stream, s, rec, out, phases);
  return check_launch("nr_debug_xcd_probe");
}

// ---- persistent sweeps (csrc/k_gru_persist.h): one launch per sweep, state exchanged inside each XCD ---------------------------------------
// The team words of the two sweeps (forward / backward) live in the module: at most ONE forward and ONE backward sweep of a process may be in
// flight at a time (they are stream-ordered in every caller of this library).  Resolved per DEVICE (a module global has one address per device).
__device__ unsigned int g_xcd_words[2][32];
// Fault words of the process (include/nr_engine.h, nr_set_fault_words): [0] / [1] sticky error bits of the forward / backward sweeps, [2] the
// step index of the first optimiser step that was skipped because of them, [3] spare.  The library's own block unless the caller attached one.
__device__ unsigned int g_fault_default[4];
static unsigned int* g_fault_user = nullptr;
static unsigned int* device_symbol(int which) {            // 0: g_xcd_words, 1: g_fault_default -- cached per device
  static unsigned int* cache[2][64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (cache[which][dev] == nullptr) {
    unsigned int* base = nullptr;
    const hipError_t e = which == 0 ? hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_xcd_words)) : hipGetSymbolAddress((void**)&base, HIP_SYMBOL(g_fault_default));
    if (e != hipSuccess) return nullptr;
    cache[which][dev] = base;
  }
  return cache[which][dev];
}
static unsigned int* xcd_words(int which) {
  unsigned int* base = device_symbol(0);
  return base ? base + which * 32 : nullptr;
}
static unsigned int* fault_words() { return g_fault_user != nullptr ? g_fault_user : device_symbol(1); }
// debug (nr_debug_gru_fault): countdown of persistent sweeps of each kind until one is launched with an injected fault (0 = off)
static int g_gru_fault_countdown[2] = {0, 0};
static int take_fault(int which) {
  if (g_gru_fault_countdown[which] <= 0) return 0;
  return --g_gru_fault_countdown[which] == 0 ? 1 : 0;
}
static long long* g_gru_stamps = nullptr;
static long long* g_gru_stamps_bwd = nullptr;
int nr_debug_gru_stamps(int64_t* buf) { g_gru_stamps = (long long*)buf; return NR_OK; }
int nr_debug_gru_stamps_bwd(int64_t* buf) { g_gru_stamps_bwd = (long long*)buf; return NR_OK; }
static int gru_persist_knob() {
  const char* e = std::getenv("NR_GRU_PERSIST");      // bit 0: forward sweep, bit 1: backward sweep (read per call)
  return e ? std::atoi(e) : 3;
}
// the persistent form needs the MI355X shape it was built for: 256 CUs (8 XCDs x 32), the reference's hidden sizes, B <= 512 (64 samples per XCD)
static bool gru_persist_ok(int B, int Hd, int T) {
  const int Hp = ceil_to(Hd + 1, 32);
  return gru_persist_knob() != 0 && T >= 2 && B >= 1 && B <= 512 && (Hp == 29 * 32 || Hp == 15 * 32) && nr::device_cus() == nr::NR_XCDS * nr::NR_XCD_TEAM;
}

static int gru_fwd_persist_launch(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, uint16_t* h_t2,
                                  uint16_t* H_all, float* h_f2, uint16_t* gates, int B, int N, int Hd, int T, void* stream) {
  if (!gi || !Whh || !b_ih || !b_hh || !len) return fail(NR_ERR_BADARG, "nr_gru_fwd_seq: bad argument");
  unsigned int* words = xcd_words(0);
  unsigned int* fw = fault_words();
  if (words == nullptr || fw == nullptr) return NR_ERR_UNSUPPORTED;
  nr::GruSeqFwdParams p;
  p.gi = gi; p.Whh = Whh; p.b_ih = b_ih; p.b_hh = b_hh; p.len = len; p.h_t2 = h_t2; p.H_all = H_all; p.h_f2 = h_f2; p.gates = gates;
  p.B = B; p.N = N; p.Hd = Hd; p.Hg = ceil_to(Hd, 16); p.Hp = ceil_to(Hd + 1, 32); p.T = T; p.sync = nr::XcdSync{words, fw, take_fault(0)}; p.stamps = g_gru_stamps;
  if (hipMemsetAsync(words, 0, 32 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess) return fail(NR_ERR_LAUNCH, "nr_gru_fwd_seq: memset");
  const int grid = nr::NR_XCDS * nr::NR_XCD_TEAM;
  if (p.Hp == 29 * 32) {
    using G = nr::GruPersistGeom<29>;
    if (allow_smem(nr::gru_fwd_persist_kernel<29>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_gru_fwd_seq: cannot reserve LDS");
    NR_LAUNCH2(nr::gru_fwd_persist_kernel<29>, grid, 1, G::NT, G::SMEM, (hipStream_t)stream, p);
  } else {
    using G = nr::GruPersistGeom<15>;
    if (allow_smem(nr::gru_fwd_persist_kernel<15>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_gru_fwd_seq: cannot reserve LDS");
    NR_LAUNCH2(nr::gru_fwd_persist_kernel<15>, grid, 1, G::NT, G::SMEM, (hipStream_t)stream, p);
  }
  return check_launch("nr_gru_fwd_seq");
}

static int gru_bwd_persist_launch(const float* g_last, const uint16_t* WhhT, const uint16_t* gates, const uint16_t* H_all, const int32_t* len, uint16_t* dgi,
                                  uint16_t* dgh, uint16_t* dgh_t2, float* carry2, int B, int N, int Hd, int T, void* stream) {
  if (!WhhT || !len) return fail(NR_ERR_BADARG, "nr_gru_bwd_seq: bad argument");
  unsigned int* words = xcd_words(1);
  unsigned int* fw = fault_words();
  if (words == nullptr || fw == nullptr) return NR_ERR_UNSUPPORTED;
  nr::GruSeqBwdParams p;
  p.g_last = g_last; p.WhhT = WhhT; p.gates = gates; p.H_all = H_all; p.len = len; p.dgi = dgi; p.dgh = dgh; p.dgh_t2 = dgh_t2; p.carry2 = carry2;
  p.B = B; p.N = N; p.Hd = Hd; p.Hg = ceil_to(Hd, 16); p.Hp = ceil_to(Hd + 1, 32); p.Kp = ceil_to(3 * p.Hg, 32); p.T = T; p.sync = nr::XcdSync{words, fw ? fw + 1 : nullptr, take_fault(1)}; p.stamps = g_gru_stamps_bwd;
  if (hipMemsetAsync(words, 0, 32 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess) return fail(NR_ERR_LAUNCH, "nr_gru_bwd_seq: memset");
  const int grid = nr::NR_XCDS * nr::NR_XCD_TEAM;
  if (p.Kp == 86 * 32) {
    using G = nr::GruBwdPersistGeom<86>;
    if (allow_smem(nr::gru_bwd_persist_kernel<86>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_gru_bwd_seq: cannot reserve LDS");
    NR_LAUNCH2(nr::gru_bwd_persist_kernel<86>, grid, 1, G::NT, G::SMEM, (hipStream_t)stream, p);
  } else if (p.Kp == 44 * 32) {
    using G = nr::GruBwdPersistGeom<44>;
    if (allow_smem(nr::gru_bwd_persist_kernel<44>, G::SMEM)) return fail(NR_ERR_LAUNCH, "nr_gru_bwd_seq: cannot reserve LDS");
    NR_LAUNCH2(nr::gru_bwd_persist_kernel<44>, grid, 1, G::NT, G::SMEM, (hipStream_t)stream, p);
  } else {
    return NR_ERR_UNSUPPORTED;
  }
  return check_launch("nr_gru_bwd_seq");
}

// STICKY error bits of the persistent sweeps since the last nr_fault_clear (0 = clean; bit 0 = a workgroup found its XCD's team full; bit 1 = a
// wait gave up): SYNCHRONISES the device.  Non-zero means: the outputs of that sweep were garbage, and every optimiser step from that one on
// was skipped (k_optim.h) -- the caller repeats them with NR_GRU_PERSIST=0 after nr_fault_clear (train_fast.py does).
int nr_fault_state(uint32_t* out4) {
  if (!out4) return fail(NR_ERR_BADARG, "nr_fault_state: null pointer");
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  unsigned int* fw = fault_words();
  if (fw == nullptr) return NR_OK;
  if (hipDeviceSynchronize() != hipSuccess) return fail(NR_ERR_LAUNCH, "nr_fault_state: device error");
  if (hipMemcpy(out4, fw, 16, hipMemcpyDeviceToHost) != hipSuccess) return fail(NR_ERR_LAUNCH, "nr_fault_state: copy");
  return NR_OK;
}
int nr_fault_clear(void) {
  unsigned int* fw = fault_words();
  if (fw == nullptr) return NR_OK;
  if (hipDeviceSynchronize() != hipSuccess || hipMemset(fw, 0, 16) != hipSuccess) return fail(NR_ERR_LAUNCH, "nr_fault_clear: device error");
  return NR_OK;
}
int nr_set_fault_words(uint32_t* words) {
  g_fault_user = (unsigned int*)words;
  return NR_OK;
}
int nr_debug_gru_fault(int which, int nth) {
  if (which < 0 || which > 1 || nth < 0) return fail(NR_ERR_BADARG, "nr_debug_gru_fault: bad argument");
  g_gru_fault_countdown[which] = nth;
  return NR_OK;
}
int nr_gru_persist_status(int32_t* fwd, int32_t* bwd) {
  uint32_t w[4];
  const int rc = nr_fault_state(w);
  if (rc != NR_OK) return rc;
  if (fwd) *fwd = (int32_t)w[0];
  if (bwd) *bwd = (int32_t)w[1];
  return NR_OK;
}
#else       // emulator build: the step-per-launch form only
int nr_debug_xcd_probe(uint32_t*, uint32_t*, uint32_t*, int, void*) { return fail(NR_ERR_UNSUPPORTED, "nr_debug_xcd_probe: not in the emulator build"); }
int nr_debug_gru_stamps(int64_t*) { return NR_OK; }
int nr_debug_gru_stamps_bwd(int64_t*) { return NR_OK; }
static bool gru_persist_ok(int, int, int) { return false; }
static int gru_persist_knob() { return 0; }
static int gru_fwd_persist_launch(const float*, const uint16_t*, const float*, const float*, const int32_t*, uint16_t*, uint16_t*, float*, uint16_t*, int, int, int,
                                  int, void*) { return NR_ERR_UNSUPPORTED; }
static int gru_bwd_persist_launch(const float*, const uint16_t*, const uint16_t*, const uint16_t*, const int32_t*, uint16_t*, uint16_t*, uint16_t*, float*, int,
                                  int, int, int, void*) { return NR_ERR_UNSUPPORTED; }
// fault words in the emulator build: host memory (the optimiser gate of k_optim.h is exercised by the CPU tests through nr_set_fault_words)
static unsigned int g_fault_default[4];
static unsigned int* g_fault_user = nullptr;
static unsigned int* fault_words() { return g_fault_user != nullptr ? g_fault_user : g_fault_default; }
int nr_fault_state(uint32_t* out4) {
  if (!out4) return fail(NR_ERR_BADARG, "nr_fault_state: null pointer");
  for (int i = 0; i < 4; ++i) out4[i] = fault_words()[i];
  return NR_OK;
}
int nr_fault_clear(void) {
  for (int i = 0; i < 4; ++i) fault_words()[i] = 0;
  return NR_OK;
}
int nr_set_fault_words(uint32_t* words) { g_fault_user = (unsigned int*)words; return NR_OK; }
int nr_debug_gru_fault(int, int) { return fail(NR_ERR_UNSUPPORTED, "nr_debug_gru_fault: not in the emulator build"); }
int nr_gru_persist_status(int32_t* fwd, int32_t* bwd) {
  if (fwd) *fwd = (int32_t)fault_words()[0];
  if (bwd) *bwd = (int32_t)fault_words()[1];
  return NR_OK;
}
#endif

int nr_gru_persist_enabled(int B, int Hd, int T) { return gru_persist_ok(B, Hd, T) ? gru_persist_knob() : 0; }

int nr_gru_seq_buffers(int B, int Hd, int T) { (void)B; (void)Hd; (void)T; return 2; }      // a ping-pong pair (the per-step buffers of the removed persistent form are gone)

int nr_gru_fwd_seq_n(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, uint16_t* h_t2, int n_buf,
                     uint16_t* H_all, float* h_f2, uint16_t* gates, int B, int N, int Hd, int T, void* stream);
int nr_gru_bwd_seq_n(const float* g_last, const uint16_t* WhhT, const uint16_t* gates, const uint16_t* H_all, const int32_t* len, uint16_t* dgi,
                     uint16_t* dgh, uint16_t* dgh_t2, int n_buf, float* carry2, int B, int N, int Hd, int T, void* stream);

int nr_gru_fwd_seq(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, uint16_t* h_t2,
                   uint16_t* H_all, float* h_f2, uint16_t* gates, int B, int N, int Hd, int T, void* stream) {
  return nr_gru_fwd_seq_n(gi, Whh, b_ih, b_hh, len, h_t2, 2, H_all, h_f2, gates, B, N, Hd, T, stream);
}

int nr_gru_fwd_seq_n(const float* gi, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len, uint16_t* h_t2, int n_buf,
                     uint16_t* H_all, float* h_f2, uint16_t* gates, int B, int N, int Hd, int T, void* stream) {
  if (!h_t2 || !h_f2 || T < 0 || T > N || B < 0 || Hd <= 0 || n_buf < 2) return fail(NR_ERR_BADARG, "nr_gru_fwd_seq: bad argument");
  const int Hg = ceil_to(Hd, 16), Hp = ceil_to(Hd + 1, 32);
  const size_t ht = (size_t)ceil_to(B, 16) * Hp, hf = (size_t)B * Hp;
  if (gru_persist_ok(B, Hd, T) && (gru_persist_knob() & 1)) {
    const int rc = gru_fwd_persist_launch(gi, Whh, b_ih, b_hh, len, h_t2, H_all, h_f2, gates, B, N, Hd, T, stream);
    if (rc != NR_ERR_UNSUPPORTED) return rc;
  }
  for (int t = 0; t < T; ++t) {
    const int rc = nr_gru_fwd_step(gi, Whh, b_ih, b_hh, len, h_t2 + (t & 1) * ht, H_all ? H_all + (size_t)(t + 1) * hf : nullptr,
                                   h_t2 + ((t + 1) & 1) * ht, h_f2 + (t & 1) * hf, h_f2 + ((t + 1) & 1) * hf,
                                   gates ? gates + (size_t)t * B * 4 * Hg : nullptr, B, N, Hd, t, stream);
    if (rc) return rc;
  }
  return NR_OK;
}

int nr_gru_fwd_seq_rows(const float* gi, const int32_t* gi_row, const uint16_t* Whh, const float* b_ih, const float* b_hh, const int32_t* len,
                        uint16_t* h_t2, float* h_f, const int32_t* active, int B, int N, int Hd, int T, void* stream) {
  if (!gi || !gi_row || !h_t2 || !h_f || T < 0 || T > N || B < 0 || Hd <= 0) return fail(NR_ERR_BADARG, "nr_gru_fwd_seq_rows: bad argument");
  const int Hp = ceil_to(Hd + 1, 32);
  const size_t ht = (size_t)ceil_to(B, 16) * Hp;
  for (int t = 0; t < T; ++t) {
    int Bt = B;
    if (active != nullptr) {
      Bt = active[t];
      if (Bt < 0 || Bt > B || (t > 0 && Bt > active[t - 1])) return fail(NR_ERR_BADARG, "nr_gru_fwd_seq_rows: active[] must be non-increasing in [0, B]");
    }
    if (Bt == 0) break;
    // the fp32 state is read and written by the same lane: one buffer, in place -- rows beyond Bt simply keep their final state
    const int rc = gru_fwd_step_launch(gi, gi_row, Whh, b_ih, b_hh, len, h_t2 + (t & 1) * ht, nullptr, h_t2 + ((t + 1) & 1) * ht, h_f, h_f, nullptr, Bt, N,
                                       Hd, t, stream);
    if (rc) return rc;
  }
  return NR_OK;
}

int nr_gru_gate_rows(const float* gi, const int32_t* gi_row, const float* gh, const float* b_ih, const float* b_hh, const int32_t* len, float* h_f,
                     uint16_t* h_b, int B, int N, int Hd, int t, void* stream) {
  if (!gi || !gi_row || !gh || !b_ih || !b_hh || !len || !h_f || !h_b || B < 0 || N <= 0 || Hd <= 0 || t < 0 || t >= N)
    return fail(NR_ERR_BADARG, "nr_gru_gate_rows: bad argument");
  if (B == 0) return NR_OK;
  nr::GruGateParams p;
  p.gi = gi; p.gi_row = gi_row; p.gh = gh; p.b_ih = b_ih; p.b_hh = b_hh; p.len = len; p.h_f = h_f; p.h_b = h_b;
  p.B = B; p.N = N; p.Hd = Hd; p.Hg = ceil_to(Hd, 16); p.Hp = ceil_to(Hd + 1, 32); p.t = t;
  NR_LAUNCH(nr::gru_gate_rows_kernel, grid_for((int64_t)B * (p.Hg / 4), 256, 16384), 256, 0, (hipStream_t)stream, p);
  return check_launch("nr_gru_gate_rows");
}

int nr_gru_bwd_seq(const float* g_last, const uint16_t* WhhT, const uint16_t* gates, const uint16_t* H_all, const int32_t* len, uint16_t* dgi,
                   uint16_t* dgh, uint16_t* dgh_t2, float* carry2, int B, int N, int Hd, int T, void* stream) {
  return nr_gru_bwd_seq_n(g_last, WhhT, gates, H_all, len, dgi, dgh, dgh_t2, 2, carry2, B, N, Hd, T, stream);
}

int nr_gru_bwd_seq_n(const float* g_last, const uint16_t* WhhT, const uint16_t* gates, const uint16_t* H_all, const int32_t* len, uint16_t* dgi,
                     uint16_t* dgh, uint16_t* dgh_t2, int n_buf, float* carry2, int B, int N, int Hd, int T, void* stream) {
  if (!g_last || !gates || !H_all || !dgi || !dgh || !dgh_t2 || !carry2 || T <= 0 || T > N || B < 0 || Hd <= 0 || n_buf < 2)
    return fail(NR_ERR_BADARG, "nr_gru_bwd_seq: bad argument");
  const int Hg = ceil_to(Hd, 16), Hp = ceil_to(Hd + 1, 32), Kp = ceil_to(3 * Hg, 32);
  const size_t dt = (size_t)ceil_to(B, 16) * Kp, cf = (size_t)B * Hp, hb = (size_t)B * Hp, gb = (size_t)B * 4 * Hg, db = (size_t)B * Kp;
  if (gru_persist_ok(B, Hd, T + 1) && (gru_persist_knob() & 2)) {
    const int rc = gru_bwd_persist_launch(g_last, WhhT, gates, H_all, len, dgi, dgh, dgh_t2, carry2, B, N, Hd, T, stream);
    if (rc != NR_ERR_UNSUPPORTED) return rc;
  }
  int i = 0;
  for (int t = T - 1; t >= -1; --t, ++i) {
    const int first = i == 0;
    const int rc = nr_gru_bwd_step(first ? g_last : nullptr, first ? nullptr : dgh_t2 + ((i + 1) & 1) * dt, first ? nullptr : carry2 + ((i + 1) & 1) * cf,
                                   WhhT, t >= 0 ? gates + (size_t)t * gb : nullptr, t >= 0 ? H_all + (size_t)t * hb : nullptr, len,
                                   t >= 0 ? dgi : nullptr, t >= 0 ? dgh + (size_t)t * db : nullptr, t >= 0 ? dgh_t2 + (i & 1) * dt : nullptr,
                                   carry2 + (i & 1) * cf, B, N, Hd, t, first, stream);
    if (rc) return rc;
  }
  return NR_OK;
}

int nr_impression_metrics(const float* scores, const int32_t* labels, const int64_t* ptr, float* out, int64_t n_impr, void* stream) {
  if (!scores || !labels || !ptr || !out || n_impr < 0) return fail(NR_ERR_BADARG, "nr_impression_metrics: bad argument");
  if (n_impr == 0) return NR_OK;
  NR_LAUNCH(nr::impression_metrics_kernel, (n_impr + 3) / 4, 256, 0, (hipStream_t)stream, scores, labels, ptr, out, n_impr);
  return check_launch("nr_impression_metrics");
}

// ---- optimiser (src/train.py:127-128,227-233) ------------------------------------------------------------------------------------
static nr::AdamCfg make_adam(const float* sched, double beta1, double beta2, double eps) {
  nr::AdamCfg c;
  c.t_dev = nullptr;
  c.gate = fault_words();                 // the optimiser never applies a step computed from a failed persistent sweep (k_optim.h)
  c.sched = sched; c.om_b1 = (float)(1.0 - beta1); c.b2 = (float)beta2; c.om_b2 = (float)(1.0 - beta2); c.eps = (float)eps;
  return c;
}
static bool bad_betas(double b1, double b2, double eps) { return !(b1 >= 0.0 && b1 < 1.0 && b2 >= 0.0 && b2 < 1.0 && eps >= 0.0); }

int nr_adam_flat(float* p, float* g, float* m, float* v, int64_t n, const float* sched, int64_t step, double beta1, double beta2, double eps,
                 float grad_scale, int zero_grad, void* stream) {
  if (!p || !g || !m || !v || !sched || n < 0 || step < 1 || bad_betas(beta1, beta2, eps)) return fail(NR_ERR_BADARG, "nr_adam_flat: bad argument");
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(NR_ERR_BADARG, "nr_adam_flat: buffers must be 16-byte aligned");
  if (n == 0) return NR_OK;
  nr::AdamCfg cfg = make_adam(sched, beta1, beta2, eps);
  cfg.t_dev = g_step_ctr;                  // with a device step counter attached, the kernel reads the step index from it
  NR_LAUNCH(nr::adam_flat_kernel, grid_for((n + 3) / 4, 256, 256 * 16), 256, 0, (hipStream_t)stream, p, g, m, v, n, cfg, step, grad_scale, zero_grad);
  return check_launch("nr_adam_flat");
}

int nr_set_step_counter(const uint32_t* ctr) {
  g_step_ctr = ctr;
  return NR_OK;
}

int nr_step_counter_add(uint32_t* ctr, uint32_t inc, void* stream) {
  if (!ctr) return fail(NR_ERR_BADARG, "nr_step_counter_add: null pointer");
  NR_LAUNCH(nr::step_counter_add_kernel, 1, 64, 0, (hipStream_t)stream, ctr, inc);
  return check_launch("nr_step_counter_add");
}

int nr_row_adam_catchup_ex(const int64_t* ids, int64_t n, float* p, float* m, float* v, int32_t* last, int64_t num_rows, int d, const float* sched,
                           int64_t upto, int by_value, double beta1, double beta2, double eps, void* stream) {
  if (!ids || !p || !m || !v || !last || !sched || n < 0 || num_rows <= 0 || d <= 0 || d > 64 * nr::ROW_EPL || upto < 0 || bad_betas(beta1, beta2, eps))
    return fail(NR_ERR_BADARG, "nr_row_adam_catchup: bad argument");
  const bool counter = g_step_ctr != nullptr && !by_value;
  if (n == 0 || (upto == 0 && !counter)) return NR_OK;
  nr::AdamCfg cfg = make_adam(sched, beta1, beta2, eps);
  cfg.t_dev = counter ? g_step_ctr : nullptr;      // inside a step with a device counter attached: `upto` is read from it (counter - 1) by the kernel
  NR_LAUNCH(nr::row_adam_catchup_kernel, (n + 3) / 4, 256, 0, (hipStream_t)stream, ids, n, p, m, v, (int*)last, num_rows, d, upto, cfg);
  return check_launch("nr_row_adam_catchup");
}

int nr_row_adam_catchup(const int64_t* ids, int64_t n, float* p, float* m, float* v, int32_t* last, int64_t num_rows, int d, const float* sched,
                        int64_t upto, double beta1, double beta2, double eps, void* stream) {
  return nr_row_adam_catchup_ex(ids, n, p, m, v, last, num_rows, d, sched, upto, 0, beta1, beta2, eps, stream);
}

int nr_row_adam_flush(float* p, float* m, float* v, int32_t* last, int64_t num_rows, int d, const float* sched, int64_t upto, double beta1,
                      double beta2, double eps, void* stream) {
  if (!p || !m || !v || !last || !sched || num_rows <= 0 || d <= 0 || d > 64 * nr::ROW_EPL || upto < 0 || bad_betas(beta1, beta2, eps))
    return fail(NR_ERR_BADARG, "nr_row_adam_flush: bad argument");
  if (upto == 0) return NR_OK;
  NR_LAUNCH(nr::row_adam_flush_kernel, grid_for(num_rows, 4, 256 * 8), 256, 0, (hipStream_t)stream, p, m, v, (int*)last, num_rows, d, upto,
            make_adam(sched, beta1, beta2, eps));
  return check_launch("nr_row_adam_flush");
}

int nr_row_adam_step(const int64_t* ids_sorted, const int64_t* perm, int64_t n, const float* rows, int64_t ld, float* p, float* m, float* v,
                     int32_t* last, int64_t num_rows, int d, const float* sched, int64_t step, double beta1, double beta2, double eps,
                     float grad_scale, int pad_row, void* stream) {
  if (!ids_sorted || !perm || !rows || !p || !m || !v || !last || !sched || n < 0 || num_rows <= 0 || d <= 0 || d > 64 * nr::ROW_EPL ||
      ld < d || step < 1 || pad_row < -1 || bad_betas(beta1, beta2, eps))
    return fail(NR_ERR_BADARG, "nr_row_adam_step: bad argument");
  if (n == 0) return NR_OK;
  nr::AdamCfg cfg = make_adam(sched, beta1, beta2, eps);
  cfg.t_dev = g_step_ctr;
  NR_LAUNCH(nr::row_adam_step_kernel, (n + 3) / 4, 256, 0, (hipStream_t)stream, ids_sorted, perm, n, rows, ld, p, m, v, (int*)last, num_rows, d,
            step, cfg, grad_scale, pad_row);
  return check_launch("nr_row_adam_step");
}

// ---- token-id sort for the embedding backward ---------------------------------------------------------------------------------------
static void sort_plan(int64_t num_rows, int* passes, int* bits) {
  int total = 1;
  while ((1LL << total) < num_rows) ++total;
  *passes = (total + 8) / 9;
  *bits = (total + *passes - 1) / *passes;
}

int64_t nr_sort_ids_workspace(int64_t n, int64_t num_rows) {
  if (n < 0 || num_rows <= 0 || n >= (1LL << 31) || num_rows > (1LL << 27)) return -1;
  int passes, bits;
  sort_plan(num_rows, &passes, &bits);
  const int64_t tiles = (n + nr::SORT_TILE - 1) / nr::SORT_TILE;
  const int64_t hist = (((int64_t)(1 << bits) * tiles * 4) + 255) / 256 * 256 + nr::SORT_MAXBINS * 4;     // tile counts + digit totals
  const int64_t pair = ((n * 4) + 255) / 256 * 256;
  return hist + 2 * pair * (passes > 2 ? 2 : 1);
}

int nr_sort_ids(const int64_t* ids, int64_t n, int64_t num_rows, int64_t* ids_sorted, int64_t* perm, void* workspace, int64_t workspace_bytes,
                void* stream) {
  const int64_t need = nr_sort_ids_workspace(n, num_rows);
  if (!ids || !ids_sorted || !perm || need < 0 || (n > 0 && (!workspace || workspace_bytes < need)) || ((uintptr_t)workspace & 15))
    return fail(NR_ERR_BADARG, "nr_sort_ids: bad argument");
  if (n == 0) return NR_OK;
  int passes, bits;
  sort_plan(num_rows, &passes, &bits);
  const int tiles = (int)((n + nr::SORT_TILE - 1) / nr::SORT_TILE);
  const int64_t hist_only = (((int64_t)(1 << bits) * tiles * 4) + 255) / 256 * 256, hist_b = hist_only + nr::SORT_MAXBINS * 4,
                pair = ((n * 4) + 255) / 256 * 256;
  unsigned char* ws = (unsigned char*)workspace;
  uint32_t* kbuf[2] = {(uint32_t*)(ws + hist_b), passes > 2 ? (uint32_t*)(ws + hist_b + 2 * pair) : nullptr};
  uint32_t* ibuf[2] = {(uint32_t*)(ws + hist_b + pair), passes > 2 ? (uint32_t*)(ws + hist_b + 3 * pair) : nullptr};
  for (int ps = 0; ps < passes; ++ps) {
    nr::SortPass s;
    s.ids = ps == 0 ? ids : nullptr;
    s.key_in = ps == 0 ? nullptr : kbuf[(ps - 1) & 1]; s.idx_in = ps == 0 ? nullptr : ibuf[(ps - 1) & 1];
    const bool last_pass = ps == passes - 1;
    s.key_out = last_pass ? nullptr : kbuf[ps & 1]; s.idx_out = last_pass ? nullptr : ibuf[ps & 1];
    s.ids_sorted = last_pass ? ids_sorted : nullptr; s.perm = last_pass ? perm : nullptr;
    s.hist = (int*)ws; s.bin_tot = (int*)(ws + hist_only); s.n = n; s.num_rows = num_rows; s.n_tiles = tiles; s.shift = ps * bits; s.bits = bits;
    NR_LAUNCH(nr::sort_hist_kernel, tiles, 256, nr::SORT_SMEM, (hipStream_t)stream, s);
    NR_LAUNCH(nr::sort_scan_kernel, 1 << bits, 256, 256 * 4, (hipStream_t)stream, s.hist, s.bin_tot, tiles);
    NR_LAUNCH(nr::sort_scatter_kernel, tiles, 256, nr::SORT_SCATTER_SMEM, (hipStream_t)stream, s);
  }
  return check_launch("nr_sort_ids");
}

int nr_dropout_mask(float* mask, int64_t n_elem, float p_drop, uint64_t seed, int site, void* stream) {
  if (!mask || n_elem < 0 || p_drop < 0.0f || p_drop >= 1.0f) return fail(NR_ERR_BADARG, "nr_dropout_mask: bad argument");
  if (n_elem == 0) return NR_OK;
  nr::DropCfg dc = make_drop(p_drop, seed);
  dc.enabled = 1;
  NR_LAUNCH(nr::dropout_mask_kernel, grid_for((n_elem + 3) / 4, 256, 2048), 256, 0, (hipStream_t)stream, mask, n_elem, dc, site);
  return check_launch("nr_dropout_mask");
}

int nr_probe_mfma(const uint16_t* A, const uint16_t* B, float* D, void* stream) {
  if (!A || !B || !D) return fail(NR_ERR_BADARG, "nr_probe_mfma: null pointer");
  NR_LAUNCH(nr::probe_mfma_kernel, 1, 64, 0, (hipStream_t)stream, A, B, D);
  return check_launch("nr_probe_mfma");
}

int nr_probe_tr16(const int32_t* offs, uint16_t* out, void* stream) {
  if (!offs || !out) return fail(NR_ERR_BADARG, "nr_probe_tr16: null pointer");
  NR_LAUNCH(nr::probe_tr16_kernel, 1, 64, 8192, (hipStream_t)stream, offs, out);
  return check_launch("nr_probe_tr16");
}

}  // extern "C"
