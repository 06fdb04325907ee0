// in_shader_w64g.hip — the in-shader ray marcher / path tracer for 64-neuron models, GENERAL instances (Nearest, Tiled levels, the transcendental activations, output activation, quantize_threshold: grid_device.h)
// (in_shader.h; one translation unit per width and kind so that the 9 shapes x 4 kernels of each build in parallel).
#include "in_shader.h"

namespace vnr {
VNR_DEFINE_IN_SHADER_WIDTH(64, true, g)
}  // namespace vnr
