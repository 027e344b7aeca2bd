// mdvt_tuning_on.hip -- the hooks of the MDVT_TUNING build (libmdvt_hip_tuning.so: tools/ and hook-driven tests only).
#include "mdvt_internal.h"
#include <stdlib.h>

namespace mdvt {

static const char* const kTuneNames[TUNE_COUNT] = { "MDVT_BLUR_ONE_PASS", "MDVT_DEBUG_SKIP", "MDVT_EDGE_INBAND", "MDVT_FORCE_GLOBAL", "MDVT_LDS_PAD", "MDVT_MESH_BAND", "MDVT_MESH_BAND3", "MDVT_MESH_CONV", "MDVT_MESH_OLD", "MDVT_MESH_TPB", "MDVT_NI_DUMP", "MDVT_NI_SKIP", "MDVT_PARAM_UPLOAD", "MDVT_POINTS_CFG", "MDVT_POINTS_NT", "MDVT_POOL_TAG", "MDVT_QUEUE_DUMP", "MDVT_RASTER_CONV_OFF", "MDVT_TELEA_BLOCKS", "MDVT_TELEA_DUMP", "MDVT_WS_CHUNK", "MDVT_WS_FRESH", "MDVT_WS_LAYOUT", "MDVT_WS_PAD", "MDVT_WS_POOL" };

const char* tuning_env(TuneKey k) { return (k >= 0 && k < TUNE_COUNT) ? getenv(kTuneNames[k]) : nullptr; }
bool tuning_build() { return true; }

}  // namespace mdvt
