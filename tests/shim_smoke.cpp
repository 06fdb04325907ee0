#define VNR_SHIM_OWN_MATH
#define VNR_SHIM_JSON_TEXT_TRANSPORT
#include "vnr_api_shim.hpp"
// Exit codes: 0 = everything ran (GPU present); 42 = all host-only checks passed and the first call that needs a GPU threw
// std::runtime_error (expected on a box without one); 11..14 = a host-only check failed.
int main() {
  // host-only part of the api.h surface -------------------------------------------------------------------------------
  {
    auto t = vnrCreateTransferFunction();
    const std::vector<vnr::vec3f> color = {{0.f, 0.f, 1.f}, {1.f, 0.f, 0.f}, {1.f, 1.f, 0.f}};
    const std::vector<vnr::vec2f> alpha = {{0.f, 0.f}, {0.5f, 0.25f}, {1.f, 1.f}};
    vnrTransferFunctionSetColor(t, color);
    vnrTransferFunctionSetAlpha(t, alpha);
    vnrTransferFunctionSetValueRange(t, vnr::range1f{0.25f, 0.75f});
    const std::vector<vnr::vec3f>& c = vnrTransferFunctionGetColor(t);   // api.h:160-162 return references
    const std::vector<vnr::vec2f>& a = vnrTransferFunctionGetAlpha(t);
    const vnr::range1f& r = vnrTransferFunctionGetValueRange(t);
    if (c.size() != 3 || a.size() != 3) return 11;
    if (c[1].x != 1.f || c[1].y != 0.f || c[2].y != 1.f || a[1].x != 0.5f || a[1].y != 0.25f) return 11;
    if (r.lower != 0.25f || r.upper != 0.75f) return 11;
  }
  if (!vnrRequireDecoding(4) || vnrRequireDecoding(5) || !vnrRequireDecoding(7) || vnrRequireDecoding(8) || vnrRequireDecoding(14) ||
      !vnrRequireDecoding(0) || !vnrRequireDecoding(13)) return 12;
  try { (void)vnrRequireDecoding(16); return 13; } catch (const std::runtime_error&) {}   // api.h:86: unknown rendering mode
  vnrRelease(nullptr);                   // declared in api.h:185, never defined there: a no-op here
  vnrMemoryQueryPrint("shim_smoke");     // api.cpp:538-552
  // part that needs a GPU ---------------------------------------------------------------------------------------------
  vnrJson cfg = vnrJson::parse(R"({"encoding":{"otype":"HashGrid"},"network":{"otype":"FullyFusedMLP","n_neurons":64}})");
  try {
    auto v = vnrCreateNeuralVolume(cfg, vnr::vec3i{8, 8, 8});
    auto r = vnrCreateRenderer(v);
    vnrRendererSetFramebufferSize(r, vnr::vec2i{16, 16});
    vnrRender(r);
    // decoding modes: fail before the first decode, work after GetNumberOfBlobs decode calls
    vnrRendererSetMode(r, 4);
    bool threw = false;
    try { vnrRender(r); } catch (const std::runtime_error&) { threw = true; }
    if (!threw) return 14;
    for (int b = 0; b < vnrNeuralVolumeGetNumberOfBlobs(v); ++b) vnrNeuralVolumeDecodeProgressive(v);
    vnrRender(r);
    try { (void)vnrSimpleVolumeGetNumberOfTimeSteps(v); return 14; } catch (const std::runtime_error&) {}   // not a simple volume
  } catch (const std::runtime_error& e) { return 42; }
  return 0;
}
