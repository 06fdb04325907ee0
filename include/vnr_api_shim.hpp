// vnr_api_shim.hpp — header-only binding that re-exposes the reference's C++ API (`/root/reference/api.h:28-188`)
// on top of the C-ABI of libvnr_amd.so (include/vnr_amd.h).
//
// An OVR app / `apps/*.cpp` of the reference includes this header INSTEAD of `api.h` and links `-lvnr_amd` instead of
// the static `instantvnr` target; call sites stay unchanged: same function names, shared_ptr handles, exceptions.
//
// Requirements on the including side (they are the reference's own, api.h:11-13):
//   * `nlohmann::json` >= 3.4 with BSON support (`json::to_bson`) as `<json/json.hpp>` or `<nlohmann/json.hpp>`
//   * vector types with public x/y/z(/w) members; the reference uses gdt's `vnr::vec3f` etc. (core/mathdef.h).
//     Define VNR_SHIM_OWN_MATH to get minimal layout-compatible types from this header instead.
#pragma once

#include <cstdio>
#include <map>
#include <mutex>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "vnr_amd.h"

#if defined(__has_include)
#if __has_include(<json/json.hpp>)
#include <json/json.hpp>
#elif __has_include(<nlohmann/json.hpp>)
#include <nlohmann/json.hpp>
#else
#include <json.hpp>
#endif
#else
#include <json/json.hpp>
#endif
// nlohmann < 3.4 has no BSON: define VNR_SHIM_JSON_TEXT_TRANSPORT to ship documents as JSON text instead (model configs
// only — params.json carries binary members and then has to be passed as a file path).

namespace vnr {
using json = nlohmann::json;

#ifdef VNR_SHIM_OWN_MATH
struct vec2i { int x, y; };
struct vec2f { float x, y; };
struct vec3i { int x, y, z; };
struct vec3f { float x, y, z; };
struct vec4f { float x, y, z, w; };
struct range1f { float lower, upper; };
// core/mathdef.h:51-83 (what api.h:11 brings in with the reference's own math): the voxel types, in the order and with the values
// of the VNR_AMD_TYPE_* enum of vnr_amd.h, and their sizes
enum ValueType {
  VALUE_TYPE_UINT8 = VNR_AMD_TYPE_UINT8, VALUE_TYPE_INT8 = VNR_AMD_TYPE_INT8, VALUE_TYPE_UINT16 = VNR_AMD_TYPE_UINT16,
  VALUE_TYPE_INT16 = VNR_AMD_TYPE_INT16, VALUE_TYPE_UINT32 = VNR_AMD_TYPE_UINT32, VALUE_TYPE_INT32 = VNR_AMD_TYPE_INT32,
  VALUE_TYPE_UINT64 = VNR_AMD_TYPE_UINT64, VALUE_TYPE_INT64 = VNR_AMD_TYPE_INT64, VALUE_TYPE_FLOAT = VNR_AMD_TYPE_FLOAT,
  VALUE_TYPE_FLOAT2 = VNR_AMD_TYPE_FLOAT2, VALUE_TYPE_FLOAT3 = VNR_AMD_TYPE_FLOAT3, VALUE_TYPE_FLOAT4 = VNR_AMD_TYPE_FLOAT4,
  VALUE_TYPE_DOUBLE = VNR_AMD_TYPE_DOUBLE,
};
inline int value_type_size(ValueType type)
{
  switch (type) {
  case VALUE_TYPE_UINT8: case VALUE_TYPE_INT8: return 1;
  case VALUE_TYPE_UINT16: case VALUE_TYPE_INT16: return 2;
  case VALUE_TYPE_UINT32: case VALUE_TYPE_INT32: case VALUE_TYPE_FLOAT: return 4;
  case VALUE_TYPE_UINT64: case VALUE_TYPE_INT64: case VALUE_TYPE_DOUBLE: return 8;
  default: return 0;
  }
}
#endif

namespace shim {
[[noreturn]] inline void fail() { throw std::runtime_error(vnrAmdGetLastError()); }  // api.cpp throws std::runtime_error
inline void check(int status) { if (status != VNR_AMD_OK) fail(); }
template <typename T> inline T* check_ptr(T* p) { if (!p) fail(); return p; }

// A json that is a string is a path (api.cpp:77-83,148-153,180-185,269-278); otherwise ship the document itself.
struct JsonArg {
  std::vector<std::uint8_t> bytes;
  std::string path;
  int format;
  JsonArg(const json& j, bool params)
  {
    if (j.is_string()) { path = j.get<std::string>(); format = params ? VNR_AMD_JSON_BSON_FILE : VNR_AMD_JSON_TEXT_FILE; }
#ifdef VNR_SHIM_JSON_TEXT_TRANSPORT
    else { const std::string t = j.dump(); bytes.assign(t.begin(), t.end()); format = VNR_AMD_JSON_TEXT; }
#else
    else { bytes = json::to_bson(j); format = VNR_AMD_JSON_BSON; }
#endif
  }
  const void* data() const { return path.empty() ? (const void*)bytes.data() : (const void*)path.c_str(); }
  size_t size() const { return path.empty() ? bytes.size() : path.size() + 1; }
};
}  // namespace shim
}  // namespace vnr

// api.h:28-32
typedef std::shared_ptr<vnrAmdVolume_t> vnrVolume;
typedef std::shared_ptr<vnrAmdRenderer_t> vnrRenderer;
typedef std::shared_ptr<vnrAmdTransferFunction_t> vnrTransferFunction;
typedef std::shared_ptr<vnrAmdCamera_t> vnrCamera;
typedef vnr::json vnrJson;
typedef vnr::ValueType vnrType;   // api.h:34 (device/device_impl.cpp:179 passes one to the volume it creates from memory)

// api.h:36-60 (same numeric values as the VNR_AMD_* enum)
enum vnrRenderMode {
  VNR_OPTIX_NO_SHADING = 0, VNR_OPTIX_GRADIENT_SHADING, VNR_OPTIX_FULL_SHADOW, VNR_OPTIX_SINGLE_SHADE_HEURISTIC,
  VNR_RAYMARCHING_NO_SHADING_DECODING, VNR_RAYMARCHING_NO_SHADING_SAMPLE_STREAMING, VNR_RAYMARCHING_NO_SHADING_IN_SHADER,
  VNR_RAYMARCHING_GRADIENT_SHADING_DECODING, VNR_RAYMARCHING_GRADIENT_SHADING_SAMPLE_STREAMING, VNR_RAYMARCHING_GRADIENT_SHADING_IN_SHADER,
  VNR_RAYMARCHING_SINGLE_SHADE_HEURISTIC_DECODING, VNR_RAYMARCHING_SINGLE_SHADE_HEURISTIC_SAMPLE_STREAMING,
  VNR_RAYMARCHING_SINGLE_SHADE_HEURISTIC_IN_SHADER, VNR_PATHTRACING_DECODING, VNR_PATHTRACING_SAMPLE_STREAMING,
  VNR_PATHTRACING_IN_SHADER, VNR_INVALID,
};

// ---- json (api.h:90-96): handled entirely by the app's nlohmann (same code as api.cpp:17-62) ------------------
#include <fstream>
#include <iomanip>
inline void vnrLoadJsonText(vnrJson& out, std::string filename)
{
  std::ifstream f(filename);
#ifndef VNR_SHIM_JSON_TEXT_TRANSPORT
  out = vnr::json::parse(f, nullptr, true, true);  // comments allowed, api.cpp:20
#else
  out = vnr::json::parse(f);
#endif
}
inline void vnrLoadJsonBinary(vnrJson& out, std::string filename)
{
  std::ifstream f(filename, std::ios::binary | std::ios::ate);
  std::streamsize size = f.tellg();
  f.seekg(0, std::ios::beg);
  std::vector<char> buffer(size);
#ifndef VNR_SHIM_JSON_TEXT_TRANSPORT
  if (f.read(buffer.data(), size)) out = vnr::json::from_bson(buffer);
#else
  (void)out; throw std::runtime_error("BSON needs nlohmann::json >= 3.4");
#endif
}
inline void vnrSaveJsonText(const vnrJson& root, std::string filename) { std::ofstream o(filename); o << std::setw(4) << root << std::endl; }
inline void vnrSaveJsonBinary(const vnrJson& root, std::string filename)
{
#ifndef VNR_SHIM_JSON_TEXT_TRANSPORT
  const auto b = vnr::json::to_bson(root);
  std::ofstream o(filename, std::ios::binary | std::ios::out);
  o.write((const char*)b.data(), b.size());
#else
  (void)root; (void)filename; throw std::runtime_error("BSON needs nlohmann::json >= 3.4");
#endif
}
inline vnrJson vnrCreateJsonText(std::string filename) { vnrJson j; vnrLoadJsonText(j, filename); return j; }
inline vnrJson vnrCreateJsonBinary(std::string filename) { vnrJson j; vnrLoadJsonBinary(j, filename); return j; }

// ---- camera (api.h:103-110) -----------------------------------------------------------------------------------
inline vnrCamera vnrCreateCamera() { return vnrCamera(vnr::shim::check_ptr(vnrAmdCreateCamera()), vnrAmdReleaseCamera); }
inline void vnrCameraSet(vnrCamera c, const vnrJson& scene)  // api.h:106; a json string is the path of a scene file (api.cpp:99-108)
{
  vnr::shim::JsonArg a(scene, false);
  vnr::shim::check(vnrAmdCameraSetFromScene(c.get(), a.data(), a.size(), a.format));
}
inline vnrCamera vnrCreateCamera(const vnrJson& scene) { vnrCamera c = vnrCreateCamera(); vnrCameraSet(c, scene); return c; }  // api.h:104
inline void vnrCameraSet(vnrCamera c, vnr::vec3f from, vnr::vec3f at, vnr::vec3f up)
{
  const float f[3] = {from.x, from.y, from.z}, a[3] = {at.x, at.y, at.z}, u[3] = {up.x, up.y, up.z};
  vnr::shim::check(vnrAmdCameraSet(c.get(), f, a, u));
}
inline vnr::vec3f vnrCameraGetPosition(vnrCamera c) { float v[3]; vnr::shim::check(vnrAmdCameraGet(c.get(), v, nullptr, nullptr, nullptr)); return vnr::vec3f{v[0], v[1], v[2]}; }
inline vnr::vec3f vnrCameraGetFocus(vnrCamera c) { float v[3]; vnr::shim::check(vnrAmdCameraGet(c.get(), nullptr, v, nullptr, nullptr)); return vnr::vec3f{v[0], v[1], v[2]}; }
inline vnr::vec3f vnrCameraGetUpVec(vnrCamera c) { float v[3]; vnr::shim::check(vnrAmdCameraGet(c.get(), nullptr, nullptr, v, nullptr)); return vnr::vec3f{v[0], v[1], v[2]}; }

// ---- volumes (api.h:117-148) -----------------------------------------------------------------------------------
// api.h:117: VIDI3D / DIVA scene document or the path of one; mode "GPU", "OUT_OF_CORE" or "NOTHING" (include/vnr_amd.h)
inline vnrVolume vnrCreateSimpleVolume(const vnrJson& scene, std::string mode, bool save_loaded_volume = false)
{
  vnr::shim::JsonArg a(scene, false);
  return vnrVolume(vnr::shim::check_ptr(vnrAmdCreateSimpleVolumeFromScene(a.data(), a.size(), a.format, mode.c_str(), save_loaded_volume)), vnrAmdReleaseVolume);
}
// AMD extension: a raw file without a scene document around it
inline vnrVolume vnrCreateSimpleVolumeFromRawFile(const std::string& filename, vnr::vec3i dims, int value_type, size_t offset = 0,
                                                  bool big_endian = false)
{
  const int d[3] = {dims.x, dims.y, dims.z};
  return vnrVolume(vnr::shim::check_ptr(vnrAmdCreateSimpleVolumeFromRawFile(filename.c_str(), d, value_type, offset, big_endian, 1.f, 0.f)), vnrAmdReleaseVolume);
}
inline vnrVolume vnrCreateNeuralVolume(const vnrJson& config, vnrVolume groundtruth, bool online_macrocell_construction = true)
{
  vnr::shim::JsonArg a(config, false);
  return vnrVolume(vnr::shim::check_ptr(vnrAmdCreateNeuralVolume(a.data(), a.size(), a.format, groundtruth.get(), online_macrocell_construction)), vnrAmdReleaseVolume);
}
inline vnrVolume vnrCreateNeuralVolume(const vnrJson& config, vnr::vec3i dims)
{
  vnr::shim::JsonArg a(config, false);
  const int d[3] = {dims.x, dims.y, dims.z};
  return vnrVolume(vnr::shim::check_ptr(vnrAmdCreateNeuralVolumeFromDims(a.data(), a.size(), a.format, d)), vnrAmdReleaseVolume);
}
inline vnrVolume vnrCreateNeuralVolume(const vnrJson& params)
{
  vnr::shim::JsonArg a(params, true);
  return vnrVolume(vnr::shim::check_ptr(vnrAmdCreateNeuralVolumeFromParams(a.data(), a.size(), a.format)), vnrAmdReleaseVolume);
}
inline void vnrNeuralVolumeSetModel(vnrVolume v, const vnrJson& config) { vnr::shim::JsonArg a(config, false); vnr::shim::check(vnrAmdNeuralVolumeSetModel(v.get(), a.data(), a.size(), a.format)); }
inline void vnrNeuralVolumeSetParams(vnrVolume v, const vnrJson& params) { vnr::shim::JsonArg a(params, true); vnr::shim::check(vnrAmdNeuralVolumeSetParams(v.get(), a.data(), a.size(), a.format)); }
inline double vnrNeuralVolumeGetPSNR(vnrVolume v, bool verbose) { return vnrAmdNeuralVolumeGetPSNR(v.get(), verbose); }
inline double vnrNeuralVolumeGetSSIM(vnrVolume v, bool verbose) { return vnrAmdNeuralVolumeGetSSIM(v.get(), verbose); }  // api.h:130
inline double vnrNeuralVolumeGetTestingLoss(vnrVolume v) { return vnrAmdNeuralVolumeGetTestingLoss(v.get()); }
inline double vnrNeuralVolumeGetTrainingLoss(vnrVolume v) { return vnrAmdNeuralVolumeGetTrainingLoss(v.get()); }
inline int vnrNeuralVolumeGetTrainingStep(vnrVolume v) { return vnrAmdNeuralVolumeGetTrainingStep(v.get()); }
inline int vnrNeuralVolumeGetNumberOfBlobs(vnrVolume v) { return vnrAmdNeuralVolumeGetNumberOfBlobs(v.get()); }
inline void vnrNeuralVolumeTrain(vnrVolume v, int steps, bool fast_mode) { vnr::shim::check(vnrAmdNeuralVolumeTrain(v.get(), steps, fast_mode)); }
inline void vnrNeuralVolumeDecodeProgressive(vnrVolume v) { vnr::shim::check(vnrAmdNeuralVolumeDecodeProgressive(v.get())); }                                   // api.h:137
inline void vnrNeuralVolumeDecodeInference(vnrVolume v, std::string filename) { vnr::shim::check(vnrAmdNeuralVolumeDecodeInference(v.get(), filename.c_str())); }  // api.h:139
inline void vnrNeuralVolumeDecodeReference(vnrVolume v, std::string filename) { vnr::shim::check(vnrAmdNeuralVolumeDecodeReference(v.get(), filename.c_str())); }  // api.h:140
inline void vnrNeuralVolumeSerializeParams(vnrVolume v, std::string filename) { vnr::shim::check(vnrAmdNeuralVolumeSerializeParamsToFile(v.get(), filename.c_str())); }
inline void vnrNeuralVolumeSerializeParams(vnrVolume v, vnrJson& params)
{
  void* b = nullptr; size_t n = 0;
  vnr::shim::check(vnrAmdNeuralVolumeSerializeParams(v.get(), &b, &n));
#ifndef VNR_SHIM_JSON_TEXT_TRANSPORT
  params = vnr::json::from_bson((const std::uint8_t*)b, (const std::uint8_t*)b + n);
#else
  (void)params;
#endif
  vnrAmdFreeHost(b);
}
inline void vnrVolumeSetClippingBox(vnrVolume v, vnr::vec3f lo, vnr::vec3f hi) { const float l[3] = {lo.x, lo.y, lo.z}, u[3] = {hi.x, hi.y, hi.z}; vnr::shim::check(vnrAmdVolumeSetClippingBox(v.get(), l, u)); }
inline void vnrVolumeSetScaling(vnrVolume v, vnr::vec3f s) { const float a[3] = {s.x, s.y, s.z}; vnr::shim::check(vnrAmdVolumeSetScaling(v.get(), a)); }
inline vnr::range1f vnrVolumeGetValueRange(vnrVolume v) { float r[2]; vnr::shim::check(vnrAmdVolumeGetValueRange(v.get(), r)); return vnr::range1f{r[0], r[1]}; }

// ---- transfer function (api.h:154-162) ---------------------------------------------------------------------------
// api.h:160-162 return references into the transfer function object.  The handle is opaque here, so the shim keeps a copy per handle
// (refreshed on every getter call); the copy lives exactly as long as the handle: the handle's deleter erases it, so an address the
// allocator hands out again never meets a stale entry.  A mutex, because the last owner of a handle may be another
// thread than the one that drives the API (the reference's apps hand frames to a UI thread).
namespace vnr { namespace shim {
struct TfnCopy { std::vector<vnr::vec3f> color; std::vector<vnr::vec2f> alpha; vnr::range1f range{0.0f, 1.0f}; };
inline std::mutex& tfn_copies_mutex() { static std::mutex m; return m; }
inline std::map<const void*, TfnCopy>& tfn_copies() { static std::map<const void*, TfnCopy> copies; return copies; }
inline void release_transfer_function(vnrAmdTransferFunction h)
{
  { std::lock_guard<std::mutex> g(tfn_copies_mutex()); tfn_copies().erase((const void*)h); }
  vnrAmdReleaseTransferFunction(h);
}
} }
inline vnrTransferFunction vnrCreateTransferFunction() { return vnrTransferFunction(vnr::shim::check_ptr(vnrAmdCreateTransferFunction()), vnr::shim::release_transfer_function); }
// api.h:155.  The reference decodes the scene's transfer function with OVR's tfn module (tfn::loadTransferFunction,
// serializer.cpp:192-193), which is not in its tree; an app that links that module passes its decoder in (colours, (position,
// alpha) pairs), and the value range comes from the scene exactly as create_scene_vidi__tfn takes it (serializer.cpp:212-256).
// Without a decoder the call throws instead of returning a made-up table.
typedef void (*vnrShimTfnDecoder)(const vnrJson& transfer_function, std::vector<vnr::vec3f>& color, std::vector<vnr::vec2f>& alpha);
inline vnrShimTfnDecoder& vnrShimTransferFunctionDecoder() { static vnrShimTfnDecoder d = nullptr; return d; }
inline vnrTransferFunction vnrCreateTransferFunction(const vnrJson& scene_or_path)
{
  vnrJson scene = scene_or_path;
  if (scene.is_string()) vnrLoadJsonText(scene, scene_or_path.get<std::string>());  // create_json_tfn (api.cpp:469-477)
  if (!vnrShimTransferFunctionDecoder())
    throw std::runtime_error("vnrCreateTransferFunction(scene): no decoder for the scene's transfer function (OVR tfn module, outside the "
                             "reference tree); install one with vnrShimTransferFunctionDecoder() or set colours / alphas explicitly");
  std::vector<vnr::vec3f> color; std::vector<vnr::vec2f> alpha;
  vnrShimTransferFunctionDecoder()(scene["view"]["volume"]["transferFunction"], color, alpha);
  if (!alpha.empty()) {  // serializer.cpp:207-208
    if (alpha.front().y < 0.01f) alpha.front().y = 0.f;
    if (alpha.back().y < 0.01f) alpha.back().y = 0.f;
  }
  vnrTransferFunction t = vnrCreateTransferFunction();
  vnr::shim::check(vnrAmdTransferFunctionSetColor(t.get(), color.empty() ? nullptr : &color[0].x, (int)color.size()));
  vnr::shim::check(vnrAmdTransferFunctionSetAlpha(t.get(), alpha.empty() ? nullptr : &alpha[0].x, (int)alpha.size()));
  vnr::shim::JsonArg a(scene, false);
  float r[2];
  const int st = vnrAmdSceneGetValueRange(a.data(), a.size(), a.format, r);
  if (st == VNR_AMD_OK) vnr::shim::check(vnrAmdTransferFunctionSetValueRange(t.get(), r[0], r[1]));
  else if (st != 2) vnr::shim::fail();
  return t;
}
inline void vnrTransferFunctionSetColor(vnrTransferFunction t, const std::vector<vnr::vec3f>& c) { vnr::shim::check(vnrAmdTransferFunctionSetColor(t.get(), c.empty() ? nullptr : &c[0].x, (int)c.size())); }
inline void vnrTransferFunctionSetAlpha(vnrTransferFunction t, const std::vector<vnr::vec2f>& a) { vnr::shim::check(vnrAmdTransferFunctionSetAlpha(t.get(), a.empty() ? nullptr : &a[0].x, (int)a.size())); }
inline void vnrTransferFunctionSetValueRange(vnrTransferFunction t, vnr::range1f r) { vnr::shim::check(vnrAmdTransferFunctionSetValueRange(t.get(), r.lower, r.upper)); }
// (the per-handle copies declared with vnrCreateTransferFunction above: a returned reference stays valid until the next getter call on the
// same handle or until the handle dies, which covers how the reference's apps use it)
namespace vnr { namespace shim {
inline TfnCopy& tfn_copy(const vnrTransferFunction& t)
{
  std::lock_guard<std::mutex> g(tfn_copies_mutex());
  TfnCopy& c = tfn_copies()[t.get()];
  int nc = 0, na = 0;
  check(vnrAmdTransferFunctionGetSizes(t.get(), &nc, &na));
  c.color.resize((size_t)nc); c.alpha.resize((size_t)na);
  float r[2] = {0.0f, 1.0f};
  check(vnrAmdTransferFunctionGet(t.get(), nc ? &c.color[0].x : nullptr, na ? &c.alpha[0].x : nullptr, r));
  c.range = vnr::range1f{r[0], r[1]};
  return c;
}
} }
inline const std::vector<vnr::vec3f>& vnrTransferFunctionGetColor(vnrTransferFunction t) { return vnr::shim::tfn_copy(t).color; }
inline const std::vector<vnr::vec2f>& vnrTransferFunctionGetAlpha(vnrTransferFunction t) { return vnr::shim::tfn_copy(t).alpha; }
inline const vnr::range1f& vnrTransferFunctionGetValueRange(vnrTransferFunction t) { return vnr::shim::tfn_copy(t).range; }

// ---- renderer (api.h:168-178) ---------------------------------------------------------------------------------------
inline vnrRenderer vnrCreateRenderer(vnrVolume v) { return vnrRenderer(vnr::shim::check_ptr(vnrAmdCreateRenderer(v.get())), vnrAmdReleaseRenderer); }
inline void vnrRendererSetFramebufferSize(vnrRenderer r, vnr::vec2i s) { vnr::shim::check(vnrAmdRendererSetFramebufferSize(r.get(), s.x, s.y)); }
inline void vnrRendererSetTransferFunction(vnrRenderer r, vnrTransferFunction t) { vnr::shim::check(vnrAmdRendererSetTransferFunction(r.get(), t.get())); }
inline void vnrRendererSetCamera(vnrRenderer r, vnrCamera c) { vnr::shim::check(vnrAmdRendererSetCamera(r.get(), c.get())); }
inline void vnrRendererSetMode(vnrRenderer r, int mode) { vnr::shim::check(vnrAmdRendererSetMode(r.get(), mode)); }
inline void vnrRendererSetDenoiser(vnrRenderer r, bool f) { vnr::shim::check(vnrAmdRendererSetDenoiser(r.get(), f)); }
inline void vnrRendererSetVolumeSamplingRate(vnrRenderer r, float v) { vnr::shim::check(vnrAmdRendererSetVolumeSamplingRate(r.get(), v)); }
inline void vnrRendererSetVolumeDensityScale(vnrRenderer r, float v) { vnr::shim::check(vnrAmdRendererSetVolumeDensityScale(r.get(), v)); }
inline void vnrRendererResetAccumulation(vnrRenderer r) { vnr::shim::check(vnrAmdRendererResetAccumulation(r.get())); }
inline void vnrRender(vnrRenderer r) { vnr::shim::check(vnrAmdRender(r.get())); }
inline vnr::vec4f* vnrRendererMapFrame(vnrRenderer r) { return (vnr::vec4f*)vnr::shim::check_ptr(vnrAmdRendererMapFrame(r.get())); }

// ---- misc (api.h:186-188) -------------------------------------------------------------------------------------------
inline void vnrMemoryQuery(size_t* used_by_renderer, size_t* used_by_tcnn) { vnrAmdMemoryQuery(used_by_renderer, used_by_tcnn); }
inline void vnrFreeTemporaryGPUMemory() { vnrAmdFreeTemporaryGPUMemory(); }
// api.cpp:538-552: "<str>: total used .., engine .., tcnn .., unknown ..".  The total is what this library allocated (the
// reference asks the driver for the device-wide figure), so "unknown" is always 0 here.
inline void vnrMemoryQueryPrint(const char* str)
{
  size_t r = 0, n = 0;
  vnrAmdMemoryQuery(&r, &n);
  auto pretty = [](size_t b) { char buf[64]; const double v = (double)b; if (b >= (1ull << 30)) snprintf(buf, sizeof buf, "%.2f GB", v / (1ull << 30)); else if (b >= (1ull << 20)) snprintf(buf, sizeof buf, "%.2f MB", v / (1ull << 20)); else snprintf(buf, sizeof buf, "%.2f KB", v / 1024.0); return std::string(buf); };
  printf("%s: total used %s, engine %s, tcnn %s, unknown %s\n", str ? str : "", pretty(r + n).c_str(), pretty(r).c_str(), pretty(n).c_str(), pretty(0).c_str());
}
// api.h:185 declares vnrRelease(void*) and api.cpp never defines it (handles are shared_ptr and release themselves)
inline void vnrRelease(void*) {}
// api.h:62-88 (inline there as well): the *_DECODING modes march a decoded dense volume
inline bool vnrRequireDecoding(int m)
{
  if (m < 0 || m > 15) throw std::runtime_error("unknown rendering mode");
  return m <= 4 || m == 7 || m == 10 || m == 13;
}
// api.h:118-119: time-varying raw volumes (one file per time step in the scene's dataSource / filename array)
inline int vnrSimpleVolumeGetNumberOfTimeSteps(vnrVolume v) { const int n = vnrAmdSimpleVolumeGetNumberOfTimeSteps(v.get()); if (n < 0) vnr::shim::fail(); return n; }
inline void vnrSimpleVolumeSetCurrentTimeStep(vnrVolume v, int time) { vnr::shim::check(vnrAmdSimpleVolumeSetCurrentTimeStep(v.get(), time)); }

// ---- isosurface (core/marching_cube.cuh:6-8; apps/batch_isosurface.cpp:70-76) ------------------------------------------------
// vnrMarchingCube(volume, isovalue, &ptr, &size, cuda): cuda = false hands out a new[]-ed host array (the application delete[]s it, as
// apps/batch_isosurface.cpp:76 does), cuda = true a device array
inline void vnrMarchingCube(vnrVolume volume, float isovalue, vnr::vec3f** ptr, size_t* size, bool cuda)
{
  float* xyz = nullptr;
  size_t n = 0;
  vnr::shim::check(vnrAmdMarchingCube(volume.get(), isovalue, &xyz, &n, cuda ? 1 : 0));
  *size = n;
  if (cuda) { *ptr = (vnr::vec3f*)xyz; return; }
  *ptr = new vnr::vec3f[n];
  for (size_t i = 0; i < n; ++i) (*ptr)[i] = vnr::vec3f{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
  vnrAmdFreeHost(xyz);
}
inline void vnrSaveTriangles(std::string filename, const vnr::vec3f* ptr, size_t size)
{
  vnr::shim::check(vnrAmdSaveTriangles(filename.c_str(), (const float*)ptr, size));
}
