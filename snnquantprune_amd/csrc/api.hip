// libsnnqp: version, thread-local error string, geometry helper.
#include "common.h"

namespace snnqp {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

}  // namespace snnqp

extern "C" {

int snnqp_version(void) { return SNNQP_VERSION; }

const char *snnqp_last_error(void) { return snnqp::g_last_error.c_str(); }

#ifndef SNNQP_BUILD_FLAGS
#define SNNQP_BUILD_FLAGS ""
#endif
const char *snnqp_build_flags(void) { return SNNQP_BUILD_FLAGS; }

int snnqp_conv_out_shape(const snnqp_conv_geom_t *g, int32_t *OH, int32_t *OW) {
  SNNQP_REQUIRE(g && OH && OW, SNNQP_EINVAL, "conv_out_shape: null argument");
  SNNQP_REQUIRE(g->stride_h > 0 && g->stride_w > 0 && g->in_dil_h > 0 &&
                    g->in_dil_w > 0 && g->k_dil_h > 0 && g->k_dil_w > 0 &&
                    g->KH > 0 && g->KW > 0,
                SNNQP_EINVAL, "conv_out_shape: non-positive stride/dilation/kernel");
  const int64_t hd = g->H > 0 ? (int64_t)(g->H - 1) * g->in_dil_h + 1 : 0;
  const int64_t wd = g->W > 0 ? (int64_t)(g->W - 1) * g->in_dil_w + 1 : 0;
  const int64_t kh = (int64_t)(g->KH - 1) * g->k_dil_h + 1;
  const int64_t kw = (int64_t)(g->KW - 1) * g->k_dil_w + 1;
  const int64_t th = hd + g->pad_h_lo + g->pad_h_hi;
  const int64_t tw = wd + g->pad_w_lo + g->pad_w_hi;
  *OH = th < kh ? 0 : (int32_t)((th - kh) / g->stride_h + 1);
  *OW = tw < kw ? 0 : (int32_t)((tw - kw) / g->stride_w + 1);
  return SNNQP_OK;
}

}  // extern "C"
