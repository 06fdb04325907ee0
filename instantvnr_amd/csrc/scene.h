// scene.h — scene JSON -> volume / camera / transfer-function descriptions.
//
// Replaces the parts of serializer.cpp that live in the reference tree: the VIDI3D layout (`dataSource[]`, `view.camera`,
// `view.volume.scalarMappingRange[Unnormalized]`; serializer.cpp:177-392) and the DIVA layout (`volume{}`; :137-176), with the
// dispatch on `version` of :423-477.  The transfer-function TABLE of a VIDI3D scene is decoded by `tfn::loadTransferFunction`
// of OVR's tfn module (serializer.cpp:192-193), which is not in the reference tree (EXTERNAL, no spec available here):
// `parse_scene_tfn` therefore returns the value range and fails loudly when asked for the table.
#pragma once

#include <string>
#include <vector>

#include "json.h"
#include "renderer.h"

namespace vnr {

struct SceneVolume {  // MultiVolume (instantvnr_types.h:40-56)
  struct File { std::string filename; size_t offset = 0; bool bigendian = false; };
  vec3i dims{0, 0, 0};
  int type = 8;
  float range_lo = 1.0f, range_hi = 0.0f;   // lo > hi: empty (compute from the data, neural_sampler.cpp:252-265)
  std::vector<File> data;                   // one file per time step
};

// "BYTE", "UNSIGNED_BYTE", ... (serializer.cpp:25-34) -> ValueType
int value_type_from_name(const std::string& name);

SceneVolume parse_scene_volume(const Json& root);                 // create_json_volume_stringify (:438-447)
bool parse_scene_camera(const Json& root, CameraData& camera);    // create_json_camera_stringify (:461-470); false: DIVA (no-op)
// create_json_tfn_stringify (:449-459): range only (see above).  Returns false when the scene gives no range.
bool parse_scene_tfn_range(const Json& root, float& lo, float& hi);

}  // namespace vnr
