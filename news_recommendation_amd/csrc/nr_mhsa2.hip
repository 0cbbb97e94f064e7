// Translation unit of the register-resident MHSA forward kernel (k_mhsa_fwd2.h), see nr_engine.hip for why it is separate.
#include <cstdlib>
#include "nr_common.h"
#include "k_mhsa_fwd2.h"
#include "k_pool2.h"

namespace nr {

int launch_mhsa_fwd2(const MhsaParams& p, hipStream_t stream) {
  using G = Mhsa2Geom;
  const int per_wg = G::TPW * G::NWAVE;
  const char* d = getenv("NR_MHSA_DEBUG");
  if (d != nullptr && atoi(d) != 0) {              // profiling: phase switches, see MhsaParams::debug
    MhsaParams q = p;
    q.debug = atoi(d);
    if (set_max_dynamic_lds((const void*)mhsa_fwd2_kernel<true>, G::SMEM)) return -1;
    NR_LAUNCH(mhsa_fwd2_kernel<true>, (p.n_seq + per_wg - 1) / per_wg, 256, G::SMEM, stream, q);
    return 0;
  }
  if (set_max_dynamic_lds((const void*)mhsa_fwd2_kernel<false>, G::SMEM)) return -1;
  NR_LAUNCH(mhsa_fwd2_kernel<false>, (p.n_seq + per_wg - 1) / per_wg, 256, G::SMEM, stream, p);
  return 0;
}

// register-resident pooling backward for titles (k_pool2.h): 16 titles per workgroup as 8 waves x 2 titles, two waves per SIMD (4 waves x 4 titles at
// one wave per SIMD measured 415 - 426 us against 362 - 368 in rounds 2 and 3 and is gone).  The default of large launches is the flat kernel
// of k_pool3.h; this one serves short launches and NR_POOL_FLAT=0.
template <typename G>
static int launch_pool2_bwd_t(const AdditiveBwdParams& p, hipStream_t stream) {
  const char* d = getenv("NR_POOL_DEBUG");         // profiling: phase switches of pool2_bwd_kernel (re-read per call)
  if (d != nullptr && atoi(d) != 0) {
    if (set_max_dynamic_lds((const void*)pool2_bwd_kernel<G, true>, G::BWD_SMEM)) return -1;
    NR_LAUNCH((pool2_bwd_kernel<G, true>), (p.n_seq + G::PER_WG - 1) / G::PER_WG, G::THREADS, G::BWD_SMEM, stream, p, atoi(d));
    return 0;
  }
  if (set_max_dynamic_lds((const void*)pool2_bwd_kernel<G, false>, G::BWD_SMEM)) return -1;
  NR_LAUNCH((pool2_bwd_kernel<G, false>), (p.n_seq + G::PER_WG - 1) / G::PER_WG, G::THREADS, G::BWD_SMEM, stream, p, 0);
  return 0;
}

int launch_pool2_bwd(const AdditiveBwdParams& p, hipStream_t stream) {
  return launch_pool2_bwd_t<Pool2Geom<20, 2, 8>>(p, stream);
}

// 50-token sequences: 4 per workgroup (one per wave, one wave per SIMD)
int launch_pool2_bwd50(const AdditiveBwdParams& p, hipStream_t stream) { return launch_pool2_bwd_t<Pool2Geom<50, 1, 4>>(p, stream); }

}  // namespace nr
