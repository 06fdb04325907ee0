// in_shader_w128.hip — the in-shader ray marcher / path tracer for 128-neuron models of the common kind
// (in_shader.h; one translation unit per width and kind so that the 9 shapes x 4 kernels of each build in parallel).
#include "in_shader.h"

namespace vnr {
VNR_DEFINE_IN_SHADER_WIDTH(128, false, )
}  // namespace vnr
