// dw_common.h -- shared declarations of the weight-gradient kernels of mlp.hip (k_dw, k_dw_tr).
#pragma once
#include "split_mfma.h"

namespace harl {

// the four waves of a workgroup as a WM x WN grid over the MT x NT output tiles (TM x TN tiles per wave)
template <int MT, int NT>
struct DwSplit {
  // (4 x 1 waves over 4 x 4 tiles: a 2 x 2 arrangement reads 12 instead of 15 fragments per k-step and wave and measured the
  // same, 0.195 vs 0.192 ms -- the LDS reads are not what the MFMA phase waits for)
  static constexpr int WM = MT >= 4 ? 4 : (MT == 2 ? 2 : 1);
  static constexpr int WN = 4 / WM;
  static constexpr int TM = MT / WM;
  static constexpr int TN = (NT + WN - 1) / WN;
};

}  // namespace harl
