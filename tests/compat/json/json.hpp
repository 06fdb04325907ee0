// tests/compat/json/json.hpp — TEST-ONLY stand-in for the `nlohmann::json` >= 3.8 an application of the reference builds with
// (api.h:11 includes <json/json.hpp>; api.cpp:23-47 uses json::to_bson / from_bson, :20 parse(f, nullptr, true, true)).
//
// This image only has nlohmann 3.1.1 (/opt/conda/include/json.hpp: no BSON, no binary values, no ignore_comments), so
// include/vnr_api_shim.hpp's production branch (no VNR_SHIM_JSON_TEXT_TRANSPORT) never went through a compiler.  This header makes it
// type-check and run: the 3.1.1 class under another namespace name, and a `nlohmann::json` derived from it that adds the newer
// members the shim uses, with their published signatures:
//     static std::vector<std::uint8_t> to_bson(const json&);
//     static json from_bson(const std::vector<char>&) / (const std::uint8_t* first, const std::uint8_t* last);
//     static json parse(std::istream&, parser_callback_t = nullptr, bool allow_exceptions = true, bool ignore_comments = false);
// A BSON binary value (params.json's "params_binary", "macrocell.data") is held as the object nlohmann prints for one,
// {"bytes": [...], "subtype": null}, since 3.1.1 has no binary type; to_bson turns that shape back into a binary (subtype 0).
// Not a product file: nothing under instantvnr_amd/ or include/ includes it.
#pragma once
#define nlohmann nlohmann_v311
#include <json.hpp>
#undef nlohmann

#include <cstdint>
#include <cstring>
#include <istream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

namespace nlohmann {

class json : public nlohmann_v311::json {
public:
  typedef nlohmann_v311::json base;
  using base::base;
  json() = default;
  json(const base& b) : base(b) {}
  json(base&& b) : base(std::move(b)) {}

  // ---- text: the 3.9 signature with ignore_comments (api.cpp:20) -----------------------------------------------------------
  static json parse(std::istream& in, const parser_callback_t cb = nullptr, bool allow_exceptions = true, bool ignore_comments = false)
  {
    std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    return parse(text, cb, allow_exceptions, ignore_comments);
  }
  static json parse(const std::string& text_in, const parser_callback_t cb = nullptr, bool allow_exceptions = true, bool ignore_comments = false)
  {
    std::string text = ignore_comments ? strip_comments(text_in) : text_in;
    return json(base::parse(text, cb, allow_exceptions));
  }
  static json parse(const char* text) { return parse(std::string(text)); }

  // ---- BSON (document / array / string / double / int32 / int64 / bool / null / binary subtype 0) --------------------------
  static std::vector<std::uint8_t> to_bson(const base& j)
  {
    if (!j.is_object()) throw std::runtime_error("to_bson: top level type must be an object");
    std::vector<std::uint8_t> out;
    document(out, j);
    return out;
  }
  static json from_bson(const std::uint8_t* first, const std::uint8_t* last)
  {
    size_t at = 0;
    return json(read_document(first, (size_t)(last - first), at, false));
  }
  static json from_bson(const std::vector<char>& v) { return from_bson((const std::uint8_t*)v.data(), (const std::uint8_t*)v.data() + v.size()); }
  static json from_bson(const std::vector<std::uint8_t>& v) { return from_bson(v.data(), v.data() + v.size()); }

private:
  static std::string strip_comments(const std::string& s)
  {
    std::string o;
    bool in_str = false;
    for (size_t i = 0; i < s.size(); ++i) {
      const char c = s[i];
      if (in_str) { o.push_back(c); if (c == '\\' && i + 1 < s.size()) o.push_back(s[++i]); else if (c == '"') in_str = false; continue; }
      if (c == '"') { in_str = true; o.push_back(c); continue; }
      if (c == '/' && i + 1 < s.size() && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') ++i; o.push_back('\n'); continue; }
      if (c == '/' && i + 1 < s.size() && s[i + 1] == '*') { i += 2; while (i + 1 < s.size() && !(s[i] == '*' && s[i + 1] == '/')) ++i; ++i; continue; }
      o.push_back(c);
    }
    return o;
  }
  template <typename T> static void put(std::vector<std::uint8_t>& o, T v) { const std::uint8_t* p = (const std::uint8_t*)&v; o.insert(o.end(), p, p + sizeof(T)); }
  static bool is_binary_shape(const base& j)
  {
    return j.is_object() && j.size() == 2 && j.count("bytes") && j.count("subtype") && j["bytes"].is_array();
  }
  static void element(std::vector<std::uint8_t>& o, const std::string& key, const base& v)
  {
    auto head = [&](std::uint8_t t) { o.push_back(t); o.insert(o.end(), key.begin(), key.end()); o.push_back(0); };
    if (is_binary_shape(v)) {
      head(0x05);
      const base& b = v["bytes"];
      put<std::int32_t>(o, (std::int32_t)b.size());
      o.push_back(0);
      for (const auto& x : b) o.push_back((std::uint8_t)x.get<unsigned>());
    } else if (v.is_object()) { head(0x03); document(o, v); }
    else if (v.is_array()) { head(0x04); document(o, v); }
    else if (v.is_string()) { head(0x02); const std::string s = v.get<std::string>(); put<std::int32_t>(o, (std::int32_t)s.size() + 1); o.insert(o.end(), s.begin(), s.end()); o.push_back(0); }
    else if (v.is_boolean()) { head(0x08); o.push_back(v.get<bool>() ? 1 : 0); }
    else if (v.is_number_float()) { head(0x01); put<double>(o, v.get<double>()); }
    else if (v.is_number_integer()) {
      const std::int64_t i = v.get<std::int64_t>();
      if (i >= INT32_MIN && i <= INT32_MAX) { head(0x10); put<std::int32_t>(o, (std::int32_t)i); } else { head(0x12); put<std::int64_t>(o, i); }
    } else if (v.is_null()) head(0x0A);
    else throw std::runtime_error("to_bson: unsupported value");
  }
  static void document(std::vector<std::uint8_t>& o, const base& j)
  {
    const size_t at = o.size();
    put<std::int32_t>(o, 0);
    if (j.is_array()) { size_t i = 0; for (const auto& v : j) element(o, std::to_string(i++), v); }
    else for (auto it = j.begin(); it != j.end(); ++it) element(o, it.key(), it.value());
    o.push_back(0);
    const std::int32_t n = (std::int32_t)(o.size() - at);
    std::memcpy(o.data() + at, &n, 4);
  }
  template <typename T> static T get_at(const std::uint8_t* d, size_t n, size_t& at)
  {
    if (at + sizeof(T) > n) throw std::runtime_error("from_bson: unexpected end of input");
    T v; std::memcpy(&v, d + at, sizeof(T)); at += sizeof(T); return v;
  }
  static base read_document(const std::uint8_t* d, size_t n, size_t& at, bool as_array)
  {
    const size_t start = at;
    const std::int32_t len = get_at<std::int32_t>(d, n, at);
    if (len < 5 || start + (size_t)len > n) throw std::runtime_error("from_bson: bad document length");
    base out = as_array ? base::array() : base::object();
    for (;;) {
      const std::uint8_t t = get_at<std::uint8_t>(d, n, at);
      if (t == 0) break;
      std::string key;
      while (at < n && d[at]) key.push_back((char)d[at++]);
      ++at;
      base v;
      switch (t) {
      case 0x01: v = get_at<double>(d, n, at); break;
      case 0x02: { const std::int32_t l = get_at<std::int32_t>(d, n, at); if (l < 1 || at + (size_t)l > n) throw std::runtime_error("from_bson: bad string"); v = std::string((const char*)d + at, (size_t)l - 1); at += (size_t)l; break; }
      case 0x03: v = read_document(d, n, at, false); break;
      case 0x04: v = read_document(d, n, at, true); break;
      case 0x05: {
        const std::int32_t l = get_at<std::int32_t>(d, n, at);
        (void)get_at<std::uint8_t>(d, n, at);
        if (l < 0 || at + (size_t)l > n) throw std::runtime_error("from_bson: bad binary");
        base bytes = base::array();
        for (std::int32_t i = 0; i < l; ++i) bytes.push_back((unsigned)d[at + (size_t)i]);
        at += (size_t)l;
        v = base::object();
        v["bytes"] = bytes;
        v["subtype"] = nullptr;
        break;
      }
      case 0x08: v = get_at<std::uint8_t>(d, n, at) != 0; break;
      case 0x0A: v = nullptr; break;
      case 0x10: v = get_at<std::int32_t>(d, n, at); break;
      case 0x12: v = get_at<std::int64_t>(d, n, at); break;
      default: throw std::runtime_error("from_bson: unsupported BSON record type");
      }
      if (as_array) out.push_back(v); else out[key] = v;
    }
    if (at != start + (size_t)len) throw std::runtime_error("from_bson: document length mismatch");
    return out;
  }
};

}  // namespace nlohmann
