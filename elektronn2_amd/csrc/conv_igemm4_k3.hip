// conv_igemm4_k3.hip -- instances of the 4x4x1-MFMA implicit-GEMM kernel, 3-wide tap rows
#include "igemm4_core.hpp"

int e2i_igemm4_launch_k3(e2_ctx* ctx, const IgemmP& p, const Igemm4Extra& x, int MG, int NT, int grid, size_t lds) {
  return igemm4_dispatch<3>(ctx, p, x, MG, NT, grid, lds);
}
int e2i_igemm4_pairs_k3(int MG, int NT) { return igemm4_pairs<3>(MG, NT); }
