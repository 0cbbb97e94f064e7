// Translation unit of the register-resident MHSA forward kernel (k_mhsa_fwd2.h), see nr_engine.hip for why it is separate.
#include "nr_common.h"
#include "k_mhsa_fwd2.h"

namespace nr {

int launch_mhsa_fwd2(const MhsaParams& p, hipStream_t stream) {
  using G = Mhsa2Geom;
  if (set_max_dynamic_lds((const void*)mhsa_fwd2_kernel, G::SMEM)) return -1;
  const int per_wg = G::TPW * G::NWAVE;
  NR_LAUNCH(mhsa_fwd2_kernel, (p.n_seq + per_wg - 1) / per_wg, 256, G::SMEM, stream, p);
  return 0;
}

}  // namespace nr
