// device_nnvolume_amd.cpp — the OVR-facing half of the "nnvolume" renderer plugin (device_nnvolume_amd.h has the other).
//
// NOT COMPILED IN THIS REPOSITORY: it needs the OVR framework's headers (ovr/renderer.h, ovr/common/dylink/ObjectFactory.h), which the
// reference tree does not vendor either.  It is written against the names the reference's own plugin uses:
//   ovr::MainRenderer with init / swap / commit / render / mapframe           /root/reference/device/device.h:13-37
//   current_scene, scene.instances[0].models[0].volume_model.{volume, transfer_function}   device_impl.cpp:113-120
//   sv.data (array with ->dims, ->type, raw data), sv.grid_spacing, sv.grid_origin          :125, 151-152
//   st.color / st.opacity / st.value_range                                                   :158-171
//   params.{fbsize, camera, tfn, path_tracing, volume_sampling_rate} with update() / ref() / get()   :219-258
//   FrameBufferData::rgba->set_data(ptr, bytes, CrossDeviceBuffer::DEVICE_CUDA)              device_impl.h:55-58
//   OVR_REGISTER_OBJECT(ovr::MainRenderer, renderer, ..., nnvolume)                          device.cpp:68
// Build inside OVR: add this file and -I<repo>/include -I<repo>/ovr_plugin -lvnr_amd to the device_nnvolume target
// (device/CMakeLists.txt:18-36) instead of device.cpp / device_impl.cpp / device_nnvolume_array.cpp, with -DVNR_HAVE_OVR.
#if defined(VNR_HAVE_OVR)

#include "device_nnvolume_amd.h"

#include "ovr/renderer.h"
#include <ovr/common/dylink/ObjectFactory.h>

#include <chrono>
#include <memory>

namespace ovr::nnvolume {

class DeviceNNVolume : public MainRenderer {
public:
  DeviceNNVolume() : MainRenderer() {}
  ~DeviceNNVolume() override = default;

  void init(int /*argc*/, const char** /*argv*/) override
  {
    const auto& scene = current_scene;
    if (scene.instances.size() != 1 || scene.instances[0].models.size() != 1) throw std::runtime_error("[nnvolume] only accept one instance with one model");
    const auto& model = scene.instances[0].models[0];
    if (model.type != scene::Model::VOLUMETRIC_MODEL || model.volume_model.volume.type != scene::Volume::STRUCTURED_REGULAR_VOLUME)
      throw std::runtime_error("[nnvolume] only accept a structured regular volume");
    const auto& st = model.volume_model.transfer_function;
    const auto& sv = model.volume_model.volume.structured_regular;
    vnr_amd_plugin::StructuredVolume v;
    v.data = sv.data->data();
    v.dims[0] = sv.data->dims.x; v.dims[1] = sv.data->dims.y; v.dims[2] = sv.data->dims.z;
    v.value_type = (int)sv.data->type;   // OVR's value types and vnr::ValueType share their numbering (device_impl.cpp:179 casts one into the other)
    v.grid_origin[0] = sv.grid_origin.x; v.grid_origin[1] = sv.grid_origin.y; v.grid_origin[2] = sv.grid_origin.z;
    v.grid_spacing[0] = sv.grid_spacing.x; v.grid_spacing[1] = sv.grid_spacing.y; v.grid_spacing[2] = sv.grid_spacing.z;
    std::vector<float> rgb(3 * st.color->size()), op(st.opacity->size());
    for (size_t i = 0; i < st.color->size(); ++i) {
      const vec4f c = st.color->data_typed<vec4f>()[i];
      rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
    for (size_t i = 0; i < op.size(); ++i) op[i] = st.opacity->data_typed<float>()[i];
    dev_.init(v, rgb.data(), (int)st.color->size(), op.data(), (int)op.size(), st.value_range.x, st.value_range.y);
    commit();
  }

  void swap() override {}

  void commit() override
  {
    if (params.fbsize.update()) { const vec2i s = params.fbsize.ref(); dev_.resize(s.x, s.y); }
    if (params.camera.update()) {
      const auto& c = params.camera.ref();
      const float from[3] = {c.from.x, c.from.y, c.from.z}, at[3] = {c.at.x, c.at.y, c.at.z}, up[3] = {c.up.x, c.up.y, c.up.z};
      dev_.set_camera(from, at, up);
    }
    if (params.tfn.update()) {
      const auto& t = params.tfn.ref();   // tfn_colors: n x rgb, tfn_alphas: n x (position, alpha) (device_impl.cpp:231-245)
      dev_.set_transfer_function(t.tfn_colors.data(), (int)(t.tfn_colors.size() / 3), t.tfn_alphas.data(), (int)(t.tfn_alphas.size() / 2),
                                 t.tfn_value_range.x, t.tfn_value_range.y);
    }
    if (params.path_tracing.update()) (void)params.path_tracing.get();   // the reference reads it into a flag its render() ignores (:248-250)
    if (params.volume_sampling_rate.update()) dev_.set_volume_sampling_rate(params.volume_sampling_rate.get());
  }

  void render() override
  {
    const auto start = std::chrono::high_resolution_clock::now();
    dev_.render();
    render_time += std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - start).count();   // device.cpp:40-47
  }

  void mapframe(FrameBufferData* fb) override
  {
    size_t bytes = 0;
    const float* pixels = dev_.mapframe(&bytes);
    // a HIP device pointer; OVR's buffer tag for "lives on the GPU the renderer runs on" is DEVICE_CUDA (device_impl.h:57)
    fb->rgba->set_data((void*)pixels, bytes, CrossDeviceBuffer::DEVICE_CUDA);
  }

private:
  vnr_amd_plugin::NNVolumeDevice dev_;
};

}  // namespace ovr::nnvolume

OVR_REGISTER_OBJECT(ovr::MainRenderer, renderer, ovr::nnvolume::DeviceNNVolume, nnvolume)

#endif  // VNR_HAVE_OVR
