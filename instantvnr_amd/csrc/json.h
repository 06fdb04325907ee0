// json.h — minimal JSON value with the two wire formats the reference uses:
//   * JSON text with // and /* */ comments (api.cpp:17-21: json::parse(file, nullptr, true, true))
//   * BSON as written by nlohmann::json::to_bson (api.cpp:23-47, core/network.cu:859-877,942-955):
//     document / array / string / double / int32 / int64 / bool / null / binary(subtype 0).
// Object keys are kept sorted (nlohmann's default object_t is std::map), so BSON output is byte-compatible.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace vnr {

class Json {
public:
  enum Type { Null, Bool, Int, Double, String, Binary, Array, Object };

  Json() : type_(Null) {}
  Json(std::nullptr_t) : type_(Null) {}
  Json(bool v) : type_(Bool), b_(v) {}
  Json(int v) : type_(Int), i_(v) {}
  Json(int64_t v) : type_(Int), i_(v) {}
  Json(uint32_t v) : type_(Int), i_((int64_t)v) {}
  Json(uint64_t v) : type_(Int), i_((int64_t)v) {}
  Json(double v) : type_(Double), d_(v) {}
  Json(float v) : type_(Double), d_((double)v) {}
  Json(const char* v) : type_(String), s_(v) {}
  Json(const std::string& v) : type_(String), s_(v) {}

  static Json object() { Json j; j.type_ = Object; return j; }
  static Json array() { Json j; j.type_ = Array; return j; }
  static Json binary(const void* data, size_t size)
  {
    Json j; j.type_ = Binary; j.s_.assign((const char*)data, size); return j;
  }

  Type type() const { return type_; }
  bool is_null() const { return type_ == Null; }
  bool is_object() const { return type_ == Object; }
  bool is_array() const { return type_ == Array; }
  bool is_string() const { return type_ == String; }
  bool is_binary() const { return type_ == Binary; }
  bool is_number() const { return type_ == Int || type_ == Double; }

  // object access
  bool contains(const std::string& key) const { return type_ == Object && obj_.count(key) != 0; }
  const Json& at(const std::string& key) const;
  Json& operator[](const std::string& key);
  Json value(const std::string& key, const Json& def) const { return contains(key) ? obj_.at(key) : def; }
  const std::map<std::string, Json>& items() const { return obj_; }

  // array access
  size_t size() const { return type_ == Array ? arr_.size() : (type_ == Object ? obj_.size() : 0); }
  const Json& at(size_t i) const;
  void push_back(const Json& v);

  // scalar access (throws std::runtime_error on a type mismatch, like nlohmann's type_error)
  bool as_bool() const;
  int64_t as_int() const;
  double as_double() const;
  float as_float() const { return (float)as_double(); }
  const std::string& as_string() const;
  const std::string& as_binary() const;

  // (de)serialisation
  static Json parse_text(const char* data, size_t size);
  static Json parse_text(const std::string& s) { return parse_text(s.data(), s.size()); }
  static Json from_bson(const uint8_t* data, size_t size);
  std::string dump(int indent = -1) const;
  std::vector<uint8_t> to_bson() const;

  static Json load_text_file(const std::string& filename);
  static Json load_bson_file(const std::string& filename);

private:
  Type type_;
  bool b_ = false;
  int64_t i_ = 0;
  double d_ = 0.0;
  std::string s_;
  std::vector<Json> arr_;
  std::map<std::string, Json> obj_;

  void dump_to(std::string& out, int indent, int depth) const;
  void bson_document(std::vector<uint8_t>& out) const;
  void bson_element(std::vector<uint8_t>& out, const std::string& key) const;
};

}  // namespace vnr
