// scene.cpp — see scene.h.  Reference: serializer.cpp:25-41 (enums), :56-134 (helpers), :137-176 (DIVA), :177-392 (VIDI3D),
// :423-477 (dispatch).
#include "scene.h"

#include <fstream>
#include <limits>
#include <stdexcept>

namespace vnr {

int value_type_from_name(const std::string& n)
{
  // NLOHMANN_JSON_SERIALIZE_ENUM (serializer.cpp:25-34); ValueType numbering as in SimpleVolume::load_host.  An unknown name
  // maps to the first pair of the table there (BYTE), which is kept.
  if (n == "BYTE") return 1;
  if (n == "UNSIGNED_BYTE") return 0;
  if (n == "SHORT") return 3;
  if (n == "UNSIGNED_SHORT") return 2;
  if (n == "INT") return 5;
  if (n == "UNSIGNED_INT") return 4;
  if (n == "FLOAT") return 8;
  if (n == "DOUBLE") return 12;
  return 1;
}

static vec3i vec3i_from_json(const Json& j)  // NLOHMANN_DEFINE_TYPE_NON_INTRUSIVE(vec3i, x, y, z)
{
  return {(int)j.at("x").as_int(), (int)j.at("y").as_int(), (int)j.at("z").as_int()};
}
static vec3f vec3f_from_json(const Json& j)
{
  return {j.at("x").as_float(), j.at("y").as_float(), j.at("z").as_float()};
}

static bool file_exists(const std::string& name)
{
  std::ifstream f(name.c_str());
  return f.good();
}

static std::string valid_filename(const Json& in, const std::string& key)  // serializer.cpp:115-134
{
  if (!in.contains(key)) throw std::runtime_error("Json key 'fileName' doesnot exist");
  const Json& js = in.at(key);
  if (js.is_array()) {
    for (size_t i = 0; i < js.size(); ++i)
      if (file_exists(js.at(i).as_string())) return js.at(i).as_string();
    throw std::runtime_error("Cannot find volume file.");
  }
  return js.as_string();
}

static bool version_is(const Json& root, const char* v) { return root.contains("version") && root.at("version").is_string() && root.at("version").as_string() == v; }

static void check_version(const Json& root)
{
  if (!root.is_object()) throw std::runtime_error("has to be a JSON object");
  if (root.contains("version") && !version_is(root, "DIVA") && !version_is(root, "VIDI3D")) throw std::runtime_error("unknown JSON configuration format");
}

// create_scene_vidi__volume / __multivolume (serializer.cpp:261-318)
static SceneVolume::File vidi_file(const Json& jsdata, vec3i* dims, int* type)
{
  if (jsdata.at("format").as_string() != "REGULAR_GRID_RAW_BINARY") throw std::runtime_error("data type unimplemented");
  SceneVolume::File f;
  f.filename = valid_filename(jsdata, "fileName");
  if (!jsdata.contains("dimensions")) throw std::runtime_error("incorrect key: dimensions");
  if (!jsdata.contains("type")) throw std::runtime_error("incorrect key: type");
  const vec3i d = vec3i_from_json(jsdata.at("dimensions"));
  const int t = value_type_from_name(jsdata.at("type").as_string());
  f.offset = jsdata.contains("offset") ? (size_t)jsdata.at("offset").as_int() : 0;
  f.bigendian = jsdata.contains("endian") && jsdata.at("endian").as_string() == "BIG_ENDIAN";
  if (dims) *dims = d;
  if (type) *type = t;
  return f;
}

static bool range_from_json(const Json& r, float& lo, float& hi)  // rangeFromJson (serializer.cpp:94-104)
{
  if (!r.contains("minimum") || !r.contains("maximum")) { lo = 0.0f; hi = 0.0f; return true; }
  lo = r.at("minimum").as_float();
  hi = r.at("maximum").as_float();
  return true;
}

// the range part of create_scene_vidi__tfn (serializer.cpp:212-256)
static bool vidi_tfn_range(const Json& jsvolume, int type, float& lo, float& hi)
{
  if (jsvolume.contains("scalarMappingRangeUnnormalized")) return range_from_json(jsvolume.at("scalarMappingRangeUnnormalized"), lo, hi);
  if (jsvolume.contains("scalarMappingRange")) {
    float x, y;
    range_from_json(jsvolume.at("scalarMappingRange"), x, y);
    float m;
    switch (type) {  // numeric_limits<T>::max() * r, evaluated in float like the reference's `T * float`
    case 0: m = (float)std::numeric_limits<uint8_t>::max(); break;
    case 1: m = (float)std::numeric_limits<int8_t>::max(); break;
    case 2: m = (float)std::numeric_limits<uint16_t>::max(); break;
    case 3: m = (float)std::numeric_limits<int16_t>::max(); break;
    case 4: m = (float)std::numeric_limits<uint32_t>::max(); break;
    case 5: m = (float)std::numeric_limits<int32_t>::max(); break;
    case 8: case 12: m = 1.0f; break;
    default: throw std::runtime_error("unknown data type");
    }
    lo = m * x; hi = m * y;
    return true;
  }
  return false;  // "calculate the volume value range ..." (left empty in the reference)
}

SceneVolume parse_scene_volume(const Json& root)
{
  check_version(root);
  SceneVolume v;
  if (version_is(root, "DIVA")) {  // create_json_volume_stringify_diva (serializer.cpp:137-168)
    const Json& c = root.at("volume");
    const Json& r = c.at("range");  // vec2f {x, y}
    v.dims = vec3i_from_json(c.at("dims"));
    v.type = value_type_from_name(c.at("type").as_string());
    v.range_lo = r.at("x").as_float();
    v.range_hi = r.at("y").as_float();
    const bool be = c.contains("bigendian") ? c.at("bigendian").as_bool() : false;
    const Json& fn = c.at("filename");
    if (fn.is_array()) {
      for (size_t i = 0; i < fn.size(); ++i) v.data.push_back({fn.at(i).as_string(), 0, be});
    } else {
      v.data.push_back({fn.as_string(), 0, be});
    }
    return v;
  }
  // create_json_volume_stringify_vidi (serializer.cpp:394-416).  The reference resizes `data` to the number of sources and
  // then push_backs sources 1.., which leaves default-constructed entries in between; here `data` is the list of sources.
  const Json& ds = root.at("dataSource");
  if (!ds.is_array()) throw std::runtime_error("'dataSource' is expected to be an array");
  if (ds.size() < 1) throw std::runtime_error("'dataSource' should contain at least one element");
  v.data.push_back(vidi_file(ds.at(0), &v.dims, &v.type));
  for (size_t i = 1; i < ds.size(); ++i) v.data.push_back(vidi_file(ds.at(i), nullptr, nullptr));
  const Json& jv = root.at("view").at("volume");
  float lo, hi;
  if (vidi_tfn_range(jv, v.type, lo, hi)) { v.range_lo = lo; v.range_hi = hi; }
  return v;
}

bool parse_scene_camera(const Json& root, CameraData& camera)
{
  check_version(root);
  if (version_is(root, "DIVA")) return false;  // "TODO" in the reference: the camera is left untouched
  // create_json_camera_stringify_vidi (serializer.cpp:418-428) + create_scene_vidi__camera (:178-187)
  const Json& jc = root.at("view").at("camera");
  CameraData c;
  c.from = vec3f_from_json(jc.at("eye"));
  c.at = vec3f_from_json(jc.at("center"));
  c.up = vec3f_from_json(jc.at("up"));
  c.fovy = jc.at("fovy").as_float();
  const Json& ds = root.at("dataSource");
  if (!ds.is_array()) throw std::runtime_error("'dataSource' is expected to be an array");
  vec3i dims;
  (void)vidi_file(ds.at(0), &dims, nullptr);
  const vec3f half = {(float)dims.x / 2.0f, (float)dims.y / 2.0f, (float)dims.z / 2.0f};
  c.at = c.at - half;
  c.from = c.from - half;
  camera = c;
  return true;
}

bool parse_scene_tfn_range(const Json& root, float& lo, float& hi)
{
  check_version(root);
  if (version_is(root, "DIVA")) return false;
  const Json& ds = root.at("dataSource");
  if (!ds.is_array()) throw std::runtime_error("'dataSource' is expected to be an array");
  if (ds.size() < 1) throw std::runtime_error("'dataSource' should contain at least one element");
  if (ds.at(0).at("format").as_string() != "REGULAR_GRID_RAW_BINARY") throw std::runtime_error("data type unimplemented");
  const int type = value_type_from_name(ds.at(0).at("type").as_string());
  return vidi_tfn_range(root.at("view").at("volume"), type, lo, hi);
}

}  // namespace vnr
