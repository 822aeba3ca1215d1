// Fused SpikingBlock(QuantDense, neuron) on int8 MFMA -- placeholder until the
// kernel lands: every request is served by the direct-form kernel.
#include "kernels.h"

namespace snnqp {

const char *dense_mfma_unsupported(int, int32_t, int32_t, const snnqp_weight_t *,
                                   const int8_t *, const snnqp_neuron_t *, int) {
  return "dense MFMA kernel not built into this library";
}

int run_dense_mfma(const void *, int64_t, int64_t, int32_t, int32_t, int32_t,
                   int32_t, const snnqp_weight_t *, const int8_t *,
                   const snnqp_bn_t *, const snnqp_neuron_t *, const float *,
                   float *, uint32_t *, hipStream_t) {
  set_error("dense MFMA kernel not built into this library");
  return SNNQP_EUNSUPPORTED;
}

}  // namespace snnqp
