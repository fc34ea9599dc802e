// conv_igemm_k4.hip -- instances of the implicit-GEMM kernel for 4-wide tap rows
#include "igemm_core.hpp"

int e2i_igemm_launch_k4(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int GU, int grid, size_t lds) {
  (void)GU;
  return igemm_dispatch<4, 1, false>(ctx, p, MT, NT, grid, lds);
}
