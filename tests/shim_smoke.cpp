#define VNR_SHIM_OWN_MATH
// Two builds (tests/test_cabi.py): with the image's nlohmann 3.1.1 (no BSON: documents cross as JSON text), and - VNR_SHIM_SMOKE_BSON -
// the shim's PRODUCTION branch (json::to_bson / from_bson, api.cpp:23-47) against tests/compat/json/json.hpp, a test-only stand-in
// for the nlohmann >= 3.8 a reference application builds with.
#ifndef VNR_SHIM_SMOKE_BSON
#define VNR_SHIM_JSON_TEXT_TRANSPORT
#endif
#include "vnr_api_shim.hpp"
#include <cstdlib>
#include <cstring>
#include <iterator>
// Exit codes: 0 = everything ran (GPU present); 42 = all host-only checks passed and the first call that needs a GPU threw
// std::runtime_error (expected on a box without one); 11..17 = a host-only check failed.
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
    // the getters' per-handle copy dies with the handle: a new transfer function at a reused address never meets a stale entry
    const void* key = t.get();
    if (vnr::shim::tfn_copies().count(key) != 1) return 11;
    t.reset();
    if (vnr::shim::tfn_copies().count(key) != 0 || !vnr::shim::tfn_copies().empty()) return 11;
    auto u = vnrCreateTransferFunction();                                  // (often the same address)
    if (!vnrTransferFunctionGetColor(u).empty() || !vnrTransferFunctionGetAlpha(u).empty()) return 11;
  }
  if (!vnrRequireDecoding(4) || vnrRequireDecoding(5) || !vnrRequireDecoding(7) || vnrRequireDecoding(8) || vnrRequireDecoding(14) ||
      !vnrRequireDecoding(0) || !vnrRequireDecoding(13)) return 12;
  try { (void)vnrRequireDecoding(16); return 13; } catch (const std::runtime_error&) {}   // api.h:86: unknown rendering mode
  {  // scene overloads (api.h:104,106,155): host-only parsing
    vnrJson scene = vnrJson::parse(R"({"dataSource":[{"format":"REGULAR_GRID_RAW_BINARY","fileName":"none.raw","dimensions":{"x":10,"y":20,"z":40},
      "type":"UNSIGNED_SHORT"}],"view":{"camera":{"eye":{"x":5,"y":10,"z":-80},"center":{"x":5,"y":10,"z":20},"up":{"x":0,"y":1,"z":0},"fovy":35},
      "volume":{"scalarMappingRange":{"minimum":0.25,"maximum":0.5},"transferFunction":{}}}})");
    auto cam = vnrCreateCamera(scene);
    const vnr::vec3f p = vnrCameraGetPosition(cam), f = vnrCameraGetFocus(cam);
    if (p.x != 0.f || p.y != 0.f || p.z != -100.f || f.z != 0.f) return 15;   // eye / center - dims / 2 (serializer.cpp:425-427)
    auto cam2 = vnrCreateCamera();
    vnrCameraSet(cam2, scene);
    if (vnrCameraGetPosition(cam2).z != -100.f) return 15;
    try { (void)vnrCreateTransferFunction(scene); return 16; } catch (const std::runtime_error&) {}   // no decoder installed
    vnrShimTransferFunctionDecoder() = [](const vnrJson&, std::vector<vnr::vec3f>& c, std::vector<vnr::vec2f>& a) {
      c = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}};
      a = {{0.f, 0.005f}, {1.f, 0.8f}};
    };
    auto t = vnrCreateTransferFunction(scene);
    if (vnrTransferFunctionGetAlpha(t)[0].y != 0.f || vnrTransferFunctionGetAlpha(t)[1].y != 0.8f) return 16;   // serializer.cpp:207-208
    const vnr::range1f& r = vnrTransferFunctionGetValueRange(t);
    if (r.lower != 65535.f * 0.25f || r.upper != 65535.f * 0.5f) return 16;   // serializer.cpp:228-231
    vnrJson bad = scene; bad["version"] = "SOMETHING";
    try { (void)vnrCreateCamera(bad); return 17; } catch (const std::runtime_error&) {}   // unknown JSON configuration format
  }
#ifdef VNR_SHIM_SMOKE_BSON
  {  // the BSON branch, host only: a params.json-like document (binary members, nested objects, int32 / int64 / double / bool / null) keeps its
     // bytes through from_bson -> to_bson, through vnrSaveJsonBinary -> vnrLoadJsonBinary, and the library's own codec reads what to_bson wrote
    const char* fixture = std::getenv("VNR_SHIM_SMOKE_FIXTURE");   // tests/golden/bson_params_like.bson (byte-checked against pymongo elsewhere)
    if (!fixture) return 20;
    vnrJson doc;
    vnrLoadJsonBinary(doc, fixture);
    if (!doc.is_object() || !doc.count("parameters") || !doc["parameters"].count("params_binary")) return 21;
    if (doc["parameters"]["params_binary"]["bytes"].size() != 2 * doc["parameters"]["n_params"].get<size_t>()) return 21;
    std::ifstream f(fixture, std::ios::binary);
    const std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const std::vector<std::uint8_t> again = vnrJson::to_bson(doc);
    if (again.size() != raw.size() || std::memcmp(again.data(), raw.data(), raw.size()) != 0) return 22;
    const std::string tmp = std::string(fixture) + ".shim_roundtrip";
    vnrSaveJsonBinary(doc, tmp);
    vnrJson back;
    vnrLoadJsonBinary(back, tmp);
    std::remove(tmp.c_str());
    if (back != doc) return 23;
    // what the shim hands the C-ABI for a json argument (JsonArg): BSON bytes the library's codec accepts and re-encodes identically
    const vnr::shim::JsonArg arg(doc, true);
    if (arg.format != VNR_AMD_JSON_BSON || arg.size() != raw.size()) return 24;
    void* out = nullptr; size_t out_n = 0;
    vnr::shim::check(vnrAmdJsonConvert(arg.data(), arg.size(), arg.format, VNR_AMD_JSON_BSON, &out, &out_n));
    const bool same = out_n == raw.size() && std::memcmp(out, raw.data(), out_n) == 0;
    vnrAmdFreeHost(out);
    if (!same) return 24;
    vnrJson text = vnrJson::parse(std::string("{ // a comment, api.cpp:20\n \"a\": 1 /* and another */ }"), nullptr, true, true);
    if (text["a"].get<int>() != 1) return 25;
  }
#endif
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
    // core/marching_cube.cuh:6-8: the isosurface of the untrained network at a value it cannot reach is empty; the calls go through
    {
      vnr::vec3f* tri = nullptr; size_t n_tri = 123;
      vnrMarchingCube(v, 1.0e9f, &tri, &n_tri, false);
      if (n_tri != 0) return 19;
      delete[] tri;
    }
#ifdef VNR_SHIM_SMOKE_BSON
    {  // api.h:143-144 + api.cpp:206-220 through BSON: serialize the volume's params into a json, create a second volume from that json
      vnrJson params;
      vnrNeuralVolumeSerializeParams(v, params);
      if (!params.count("parameters") || !params.count("model") || params["volume"]["dims"]["x"].get<int>() != 8) return 26;
      auto v2 = vnrCreateNeuralVolume(params);
      if (vnrNeuralVolumeGetNumberOfBlobs(v2) != vnrNeuralVolumeGetNumberOfBlobs(v)) return 26;
      vnrJson params2;
      vnrNeuralVolumeSerializeParams(v2, params2);
      if (params2["parameters"] != params["parameters"]) return 27;      // the same weights, bit for bit
    }
#endif
    // api.h:34 vnrType = vnr::ValueType, the voxel type of a volume made from memory (device/device_impl.cpp:175-184)
    const vnrType ty = vnr::VALUE_TYPE_UINT16;
    if (vnr::value_type_size(ty) != 2 || (int)vnr::VALUE_TYPE_FLOAT != 8 || (int)vnr::VALUE_TYPE_DOUBLE != 12) return 18;
  } catch (const std::runtime_error& e) { return 42; }
  return 0;
}
