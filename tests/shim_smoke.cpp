#define VNR_SHIM_OWN_MATH
#define VNR_SHIM_JSON_TEXT_TRANSPORT
#include "vnr_api_shim.hpp"
int main() {
  vnrJson cfg = vnrJson::parse(R"({"encoding":{"otype":"HashGrid"},"network":{"otype":"FullyFusedMLP","n_neurons":64}})");
  try {
    auto v = vnrCreateNeuralVolume(cfg, vnr::vec3i{8, 8, 8});
    auto r = vnrCreateRenderer(v);
    vnrRendererSetFramebufferSize(r, vnr::vec2i{16, 16});
    vnrRender(r);
  } catch (const std::runtime_error& e) { return 42; }
  return 0;
}
