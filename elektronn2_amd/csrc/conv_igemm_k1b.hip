// conv_igemm_k1b.hip -- bf16-operand instances of the implicit-GEMM kernel for 1-wide tap
// rows (igemm_core.hpp, "bf16 operand form").
#include "igemm_core.hpp"

int e2i_igemm_launch_k1_bf(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int GU, int grid, size_t lds) {
  if (GU == 4) return igemm_dispatch<1, 4, true>(ctx, p, MT, NT, grid, lds);
  return igemm_dispatch<1, 1, true>(ctx, p, MT, NT, grid, lds);
}
