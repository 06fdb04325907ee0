// network_infer_w128.hip — instances of the fused evaluation kernel (infer_kernel.h) for FullyFusedMLP n_neurons = 128
// (tcnn_impl.cu:315-347 instantiates WIDTH 16 / 32 / 64 / 128): inference and the training forward pass, every encoding shape, for the
// common kind of model (Hash / Dense grid, Linear / Smoothstep, ReLU / None); network_infer_w128g.hip holds the GENERAL instances.
#include "infer_kernel.h"

namespace vnr {
VNR_DEFINE_FUSED_WIDTH(128, false, )
}  // namespace vnr
