// conv1d_s2.hip -- the stride-2 instantiations of the implicit-GEMM Conv1d kernels (conv1d_kernels.hpp, AMAX = 320).
// Reference: layers/conv_1d.c:128-140 (`input_row_offset = x * stride`).  Stride 2 is the ordinary sub-sampling front
// end; its 128-position tile spans (128 - 1) * 2 + k input rows, past the 192-row register-staging budget of the stride-1
// instantiations -- and widening THAT budget cost the stride-1 shapes 4-12 % (two extra, mostly idle load passes), so the
// wide window is a set of instantiations of its own, compiled in its own translation unit.
#include "conv1d_kernels.hpp"

// stride 2 (the ordinary sub-sampling front end, conv_1d.c:128-140 `input_row_offset = x * stride`): 320-row window budget, quad epilogue
template <int WM, int WN, int TM, int TN>
static int launch_mfma_s2(const ConvParams &p, bool a4, bool split) {
    if (split) return a4 ? launch_mfma_o<WM, WN, TM, TN, true, true, true, 320>(p) : launch_mfma_o<WM, WN, TM, TN, false, true, true, 320>(p);
    return a4 ? launch_mfma_o<WM, WN, TM, TN, true, false, true, 320>(p) : launch_mfma_o<WM, WN, TM, TN, false, false, true, 320>(p);
}


int nntk_conv1d_launch_s2(const ConvParams &p, bool a4, bool split) {
    if (p.Cout_p % 128 == 0) return launch_mfma_s2<2, 2, 2, 2>(p, a4, split);
    if (p.Cout_p % 64 == 0)  return launch_mfma_s2<4, 1, 1, 2>(p, a4, split);
    return launch_mfma_s2<4, 1, 1, 1>(p, a4, split);
}
