// device_nnvolume_amd.h — the OVR renderer plugin "nnvolume" on top of libvnr_amd.so.
//
// Replaces /root/reference/device/{device.h, device.cpp, device_impl.h, device_impl.cpp}: an `ovr::MainRenderer` whose
// init / swap / commit / render / mapframe (device.h:23-29) drive the sample-streaming renderer (rendering mode 5) on the scene's
// structured regular volume and hand OVR a DEVICE framebuffer (device_impl.h:55-58).
//
// Two layers, because the OVR headers (ovr/renderer.h, ovr/common/dylink/ObjectFactory.h, ...) are not in the reference tree:
//   * NNVolumeDevice (this header, plain C++ over include/vnr_amd.h): everything the reference's Impl does, with the scene handed over as
//     plain data.  Compiled and run here (tests/ovr_plugin_host.cpp, tests/test_gpu_ovr_plugin.py).
//   * ovr::nnvolume::DeviceNNVolume (device_nnvolume_amd.cpp, behind VNR_HAVE_OVR): the ~60 lines that unpack ovr::Scene / the
//     parameter block into NNVolumeDevice calls and register the plugin.  NOT compiled here; written against the member names the
//     reference's own plugin uses (device_impl.cpp:100-191, 219-258).
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "vnr_amd.h"

namespace vnr_amd_plugin {

struct StructuredVolume {        // scene::Volume::structured_regular (device_impl.cpp:119)
  const void* data = nullptr;    // host voxels, x fastest
  int dims[3] = {0, 0, 0};
  int value_type = VNR_AMD_TYPE_FLOAT;   // vnrType / vnr::ValueType: the same numbering as OVR's value types (device_impl.cpp:179)
  float grid_origin[3] = {0, 0, 0};
  float grid_spacing[3] = {1, 1, 1};
};

class NNVolumeDevice {
public:
  NNVolumeDevice() = default;
  NNVolumeDevice(const NNVolumeDevice&) = delete;
  NNVolumeDevice& operator=(const NNVolumeDevice&) = delete;
  ~NNVolumeDevice()
  {
    if (renderer_) vnrAmdReleaseRenderer(renderer_);
    if (camera_) vnrAmdReleaseCamera(camera_);
    if (tfn_) vnrAmdReleaseTransferFunction(tfn_);
    if (volume_) vnrAmdReleaseVolume(volume_);
  }

  // DeviceNNVolume::Impl::init (device_impl.cpp:100-191): volume texture, object -> world map, macrocell, transfer function, renderer in
  // rendering mode 5 with a device framebuffer.  colors: n x rgb; alphas: n opacities at the nodes i / (n - 1) (:158-171).
  void init(const StructuredVolume& v, const float* colors_rgb, int n_colors, const float* opacities, int n_opacities, float range_lo, float range_hi)
  {
    if (renderer_) throw std::runtime_error("[nnvolume] device already initialized!");   // :103-105
    // CreateArray3DScalarCUDA + MainRenderer::set_scene: the voxels, min / max normalised like every volume of the library
    // (range_lo > range_hi: computed from the data, which is what the reference's array helper reports as lower / upper)
    volume_ = check_ptr(vnrAmdCreateSimpleVolumeFromMemory(v.data, v.dims, v.value_type, 1.0f, 0.0f));
    // matrix = translate(grid_origin) * scale(grid_spacing * dims) (:151-153)
    const float m[12] = {v.grid_spacing[0] * (float)v.dims[0], 0, 0, 0, v.grid_spacing[1] * (float)v.dims[1], 0, 0, 0, v.grid_spacing[2] * (float)v.dims[2],
                         v.grid_origin[0], v.grid_origin[1], v.grid_origin[2]};
    check(vnrAmdVolumeSetTransform(volume_, m));
    tfn_ = check_ptr(vnrAmdCreateTransferFunction());
    camera_ = check_ptr(vnrAmdCreateCamera());
    renderer_ = check_ptr(vnrAmdCreateRenderer(volume_));        // (the macrocell came with the volume: sampler.cu:5-17)
    set_transfer_function_nodes(colors_rgb, n_colors, opacities, n_opacities, range_lo, range_hi);
    check(vnrAmdRendererSetMode(renderer_, 5));                  // renderer.set_rendering_mode(5) (:186)
    check(vnrAmdRendererSetOutputAsDeviceFramebuffer(renderer_, 1));   // renderer.set_output_as_cuda_framebuffer() (:187)
  }

  // DeviceNNVolume::Impl::commit (device_impl.cpp:219-258): only what changed
  void resize(int w, int h) { check(vnrAmdRendererSetFramebufferSize(renderer_, w, h)); width_ = w; height_ = h; }
  void set_camera(const float from[3], const float at[3], const float up[3])   // vnr::Camera{from, at, up}: the fovy stays the default 60 (device_impl.h:68)
  {
    check(vnrAmdCameraSet(camera_, from, at, up));
    check(vnrAmdRendererSetCamera(renderer_, camera_));
  }
  // parent->params.tfn (:229-246): colors n x rgb, alphas n x (position, alpha)
  void set_transfer_function(const float* colors_rgb, int n_colors, const float* alphas_xy, int n_alphas, float range_lo, float range_hi)
  {
    check(vnrAmdTransferFunctionSetColor(tfn_, colors_rgb, n_colors));
    check(vnrAmdTransferFunctionSetAlpha(tfn_, alphas_xy, n_alphas));
    // the scene gives the range in DATA units (the reference's plugin samples the raw texture); the library's voxels are normalised
    // to [0, 1] by the data's own min / max, so the range is mapped the same way
    float data[2] = {0.0f, 1.0f};
    check(vnrAmdSimpleVolumeGetDataRange(volume_, data));
    const float w = data[1] > data[0] ? 1.0f / (data[1] - data[0]) : 1.0f;
    check(vnrAmdTransferFunctionSetValueRange(tfn_, (range_lo - data[0]) * w, (range_hi - data[0]) * w));
    check(vnrAmdRendererSetTransferFunction(renderer_, tfn_));   // also refreshes the macrocell's max opacity (device_impl.h:72-75)
  }
  void set_volume_sampling_rate(float r) { check(vnrAmdRendererSetVolumeSamplingRate(renderer_, r)); }
  void set_volume_density_scale(float s) { check(vnrAmdRendererSetVolumeDensityScale(renderer_, s)); }
  void set_scene_clipbox(const float lower[3], const float upper[3]) { check(vnrAmdVolumeSetClippingBox(volume_, lower, upper)); }

  void render() { check(vnrAmdRender(renderer_)); }              // Impl::render (:38-97: renderer.render())
  // Impl::mapframe (device_impl.h:55-58): DEVICE pixels, width x height vec4f, valid until two frames later
  const float* mapframe(size_t* bytes)
  {
    const float* p = vnrAmdRendererMapFrame(renderer_);
    if (!p) fail();
    if (bytes) *bytes = (size_t)width_ * (size_t)height_ * 4 * sizeof(float);
    return p;
  }
  int width() const { return width_; }
  int height() const { return height_; }

private:
  void set_transfer_function_nodes(const float* colors_rgb, int n_colors, const float* opacities, int n, float lo, float hi)
  {
    std::vector<float> xy((size_t)2 * (size_t)n);
    for (int i = 0; i < n; ++i) { xy[2 * i] = n > 1 ? (float)i / (float)(n - 1) : 0.0f; xy[2 * i + 1] = opacities[i]; }   // :166-169
    set_transfer_function(colors_rgb, n_colors, xy.data(), n, lo, hi);
  }
  [[noreturn]] static void fail() { throw std::runtime_error(std::string("[nnvolume] ") + vnrAmdGetLastError()); }
  static void check(int status) { if (status != VNR_AMD_OK) fail(); }
  template <typename T> static T check_ptr(T p) { if (!p) fail(); return p; }

  vnrAmdVolume volume_ = nullptr;
  vnrAmdTransferFunction tfn_ = nullptr;
  vnrAmdCamera camera_ = nullptr;
  vnrAmdRenderer renderer_ = nullptr;
  int width_ = 0, height_ = 0;
};

}  // namespace vnr_amd_plugin
