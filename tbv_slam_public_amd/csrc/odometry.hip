// odometry.hip -- placeholder, replaced below in this round.
#include "common.hpp"
extern "C" int cfear_odometry_create(cfear_ctx* ctx, int32_t, const cfear_polar_desc*, const cfear_odometry_params*, cfear_odometry** out) {
  if (out) *out = nullptr;
  return cfear_set_error(ctx, CFEAR_ERR_INVALID_ARGUMENT, "odometry pipeline not built yet");
}
extern "C" int cfear_odometry_process(cfear_odometry*, const uint8_t*, cfear_frame_info*) { return CFEAR_ERR_INVALID_ARGUMENT; }
extern "C" int cfear_odometry_destroy(cfear_odometry*) { return CFEAR_OK; }
