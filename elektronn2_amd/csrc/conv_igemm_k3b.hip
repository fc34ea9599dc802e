// conv_igemm_k3b.hip -- bf16-operand instances of the implicit-GEMM kernel for 3-wide tap
// rows (igemm_core.hpp, "bf16 operand form"); a separate translation unit so that it
// compiles next to the f32 one.
#include "igemm_core.hpp"

int e2i_igemm_launch_k3_bf(e2_ctx* ctx, const IgemmP& p, int MT, int NT, int GU, int grid, size_t lds) {
  (void)GU;
  return igemm_dispatch<3, 1, true>(ctx, p, MT, NT, grid, lds);
}
