#pragma once
#include "rtj_common.h"

namespace mirtj {
void build_qtab(int Q, QTab* t);      // Q <= 0 gives the all-zero row
void build_all_qtabs(QTab* lut);      // lut[kNumQTab]
}  // namespace mirtj
