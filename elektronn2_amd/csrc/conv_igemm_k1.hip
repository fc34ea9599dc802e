// conv_igemm_k1.hip -- instances of the implicit-GEMM kernel for 1-wide tap rows.
// GU = 4: 1x1 taps (plain GEMM: the last conv layers, UpConv) process four channel
// groups per pipeline step so that a step carries enough MFMAs to hide its loads.
#include "igemm_core.hpp"

int e2i_igemm_launch_k1(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int GU, int grid, size_t lds) {
  if (GU == 4) return igemm_dispatch<1, 4, false>(ctx, p, MT, NT, grid, lds);
  return igemm_dispatch<1, 1, false>(ctx, p, MT, NT, grid, lds);
}
