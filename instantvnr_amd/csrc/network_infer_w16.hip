// network_infer_w16.hip — instances of the fused evaluation kernel (infer_kernel.h) for FullyFusedMLP n_neurons = 16
// (tcnn_impl.cu:315-347 instantiates WIDTH 16 / 32 / 64 / 128): inference and the training forward pass, every encoding shape, for the
// common kind of model (Hash / Dense grid, Linear / Smoothstep, ReLU / None); network_infer_w16g.hip holds the GENERAL instances.
#include "infer_kernel.h"

namespace vnr {
VNR_DEFINE_FUSED_WIDTH(16, false, )
}  // namespace vnr
