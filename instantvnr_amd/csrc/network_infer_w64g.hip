// network_infer_w64g.hip — the GENERAL instances of the fused evaluation kernel for n_neurons = 64: Tiled grids, Nearest interpolation,
// the activations Exponential / Sigmoid / Squareplus / Softplus and output activations (tcnn_impl.cu:405-415, tcnn_impl_decoder.cu:68-94).
#include "infer_kernel.h"

namespace vnr {
VNR_DEFINE_FUSED_WIDTH(64, true, g)
}  // namespace vnr
