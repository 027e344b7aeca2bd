// mdvt_tuning_off.hip -- the product library has no tuning / ablation hooks (mdvt_internal.h).
#include "mdvt_internal.h"

namespace mdvt {

const char* tuning_env(TuneKey) { return nullptr; }
bool tuning_build() { return false; }

}  // namespace mdvt
