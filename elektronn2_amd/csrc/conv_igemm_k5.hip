// conv_igemm_k5.hip -- instances of the implicit-GEMM kernel for 5-wide tap rows
#include "igemm_core.hpp"

int e2i_igemm_launch_k5(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int GU, int grid, size_t lds) {
  (void)GU;
  return igemm_dispatch<5, 1, false>(ctx, p, MT, NT, grid, lds);
}
