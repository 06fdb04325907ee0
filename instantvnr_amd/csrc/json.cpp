// json.cpp — JSON text (with comments) + BSON codec.  See json.h.
#include "json.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace vnr {


const Json& Json::at(const std::string& key) const
{
  if (type_ != Object) throw std::runtime_error("json: not an object (key '" + key + "')");
  auto it = obj_.find(key);
  if (it == obj_.end()) throw std::runtime_error("json: key '" + key + "' not found");
  return it->second;
}

Json& Json::operator[](const std::string& key)
{
  if (type_ == Null) type_ = Object;
  if (type_ != Object) throw std::runtime_error("json: not an object (key '" + key + "')");
  return obj_[key];
}

const Json& Json::at(size_t i) const
{
  if (type_ != Array || i >= arr_.size()) throw std::runtime_error("json: array index out of range");
  return arr_[i];
}

void Json::push_back(const Json& v)
{
  if (type_ == Null) type_ = Array;
  if (type_ != Array) throw std::runtime_error("json: not an array");
  arr_.push_back(v);
}

bool Json::as_bool() const
{
  if (type_ == Bool) return b_;
  throw std::runtime_error("json: type must be boolean");
}
int64_t Json::as_int() const
{
  if (type_ == Int) return i_;
  if (type_ == Double) {   // (a conversion of a value that an int64 cannot hold is undefined: 1e30 for a dimension is an error, not INT64_MIN)
    if (!(d_ > -9.2e18 && d_ < 9.2e18)) throw std::runtime_error("json: number out of range for an integer");
    return (int64_t)d_;
  }
  if (type_ == Bool) return b_ ? 1 : 0;
  throw std::runtime_error("json: type must be number");
}
double Json::as_double() const
{
  if (type_ == Double) return d_;
  if (type_ == Int) return (double)i_;
  throw std::runtime_error("json: type must be number");
}
const std::string& Json::as_string() const
{
  if (type_ == String) return s_;
  throw std::runtime_error("json: type must be string");
}
const std::string& Json::as_binary() const
{
  if (type_ == Binary) return s_;
  throw std::runtime_error("json: type must be binary");
}

// ------------------------------------------------------------------------------------------ text
namespace {
constexpr int kTextMaxDepth = 64;   // (= kBsonMaxDepth) the reference's files nest 4 deep; a crafted or damaged text must not recurse the stack away

struct Parser {
  const char* p;
  const char* end;
  int depth = 0;

  [[noreturn]] void fail(const char* msg) const { throw std::runtime_error(std::string("json parse error: ") + msg); }

  void skip_ws()
  {
    for (;;) {
      while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
      if (p + 1 < end && p[0] == '/' && p[1] == '/') {
        while (p < end && *p != '\n') ++p;
      } else if (p + 1 < end && p[0] == '/' && p[1] == '*') {
        p += 2;
        while (p + 1 < end && !(p[0] == '*' && p[1] == '/')) ++p;
        if (p + 1 >= end) fail("unterminated comment");
        p += 2;
      } else {
        return;
      }
    }
  }

  static void append_utf8(std::string& s, uint32_t cp)
  {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
      s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    }
  }

  uint32_t hex4()
  {
    if (end - p < 4) fail("bad \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = *p++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else fail("bad \\u escape");
    }
    return v;
  }

  std::string string()
  {
    if (p >= end || *p != '"') fail("expected string");
    ++p;
    std::string s;
    while (p < end && *p != '"') {
      char c = *p++;
      if (c == '\\') {
        if (p >= end) fail("bad escape");
        c = *p++;
        switch (c) {
        case '"': s.push_back('"'); break;
        case '\\': s.push_back('\\'); break;
        case '/': s.push_back('/'); break;
        case 'b': s.push_back('\b'); break;
        case 'f': s.push_back('\f'); break;
        case 'n': s.push_back('\n'); break;
        case 'r': s.push_back('\r'); break;
        case 't': s.push_back('\t'); break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp <= 0xDBFF) {   // a high surrogate needs its low half (nlohmann::json refuses a lone one too)
            if (end - p < 6 || p[0] != '\\' || p[1] != 'u') fail("lone surrogate in \\u escape");
            p += 2;
            const uint32_t lo = hex4();
            if (lo < 0xDC00 || lo > 0xDFFF) fail("bad surrogate pair in \\u escape");
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
            fail("lone surrogate in \\u escape");
          }
          append_utf8(s, cp);
          break;
        }
        default: fail("bad escape");
        }
      } else {
        s.push_back(c);
      }
    }
    if (p >= end) fail("unterminated string");
    ++p;
    return s;
  }

  Json number()
  {
    const char* b = p;
    bool is_float = false;
    if (p < end && (*p == '-' || *p == '+')) ++p;
    while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
      if (*p == '.' || *p == 'e' || *p == 'E') is_float = true;
      ++p;
    }
    const std::string tok(b, p);
    if (tok.empty()) fail("expected a value");
    // the whole token must be a number: "1-2", "--3", "1e", "." are errors, not the number in front of the damage
    char* stop = nullptr;
    if (is_float) {
      const double d = std::strtod(tok.c_str(), &stop);
      if (stop != tok.c_str() + tok.size() || stop == tok.c_str()) fail("malformed number");
      return Json(d);
    }
    errno = 0;
    const long long v = std::strtoll(tok.c_str(), &stop, 10);
    if (stop != tok.c_str() + tok.size() || stop == tok.c_str()) fail("malformed number");
    if (errno == ERANGE) return Json(std::strtod(tok.c_str(), nullptr));
    return Json((int64_t)v);
  }

  Json value()
  {
    skip_ws();
    if (p >= end) fail("unexpected end");
    const char c = *p;
    if (c == '{') {
      ++p;
      if (++depth > kTextMaxDepth) fail("nested too deeply");
      Json o = Json::object();
      skip_ws();
      if (p < end && *p == '}') { ++p; --depth; return o; }
      for (;;) {
        skip_ws();
        const std::string k = string();
        skip_ws();
        if (p >= end || *p != ':') fail("expected ':'");
        ++p;
        o[k] = value();
        skip_ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; --depth; return o; }
        fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      ++p;
      if (++depth > kTextMaxDepth) fail("nested too deeply");
      Json a = Json::array();
      skip_ws();
      if (p < end && *p == ']') { ++p; --depth; return a; }
      for (;;) {
        a.push_back(value());
        skip_ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; --depth; return a; }
        fail("expected ',' or ']'");
      }
    }
    if (c == '"') return Json(string());
    if (end - p >= 4 && !std::strncmp(p, "true", 4)) { p += 4; return Json(true); }
    if (end - p >= 5 && !std::strncmp(p, "false", 5)) { p += 5; return Json(false); }
    if (end - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; return Json(); }
    return number();
  }
};

void dump_string(std::string& out, const std::string& s)
{
  out.push_back('"');
  for (unsigned char c : s) {
    switch (c) {
    case '"': out += "\\\""; break;
    case '\\': out += "\\\\"; break;
    case '\n': out += "\\n"; break;
    case '\r': out += "\\r"; break;
    case '\t': out += "\\t"; break;
    case '\b': out += "\\b"; break;
    case '\f': out += "\\f"; break;
    default:
      if (c < 0x20) { char buf[8]; std::snprintf(buf, sizeof buf, "\\u%04x", c); out += buf; }
      else out.push_back((char)c);
    }
  }
  out.push_back('"');
}
}  // namespace

Json Json::parse_text(const char* data, size_t size)
{
  Parser ps{data, data + size};
  Json v = ps.value();
  ps.skip_ws();
  if (ps.p != ps.end) ps.fail("trailing characters");
  return v;
}

void Json::dump_to(std::string& out, int indent, int depth) const
{
  auto nl = [&](int d) {
    if (indent >= 0) { out.push_back('\n'); out.append((size_t)(indent * d), ' '); }
  };
  switch (type_) {
  case Null: out += "null"; break;
  case Bool: out += b_ ? "true" : "false"; break;
  case Int: out += std::to_string(i_); break;
  case Double: {
    if (!std::isfinite(d_)) { out += "null"; break; }
    char buf[40];
    std::snprintf(buf, sizeof buf, "%.17g", d_);
    // shortest representation that round-trips
    for (int prec = 1; prec < 17; ++prec) {
      char b2[40];
      std::snprintf(b2, sizeof b2, "%.*g", prec, d_);
      if (std::strtod(b2, nullptr) == d_) { std::strcpy(buf, b2); break; }
    }
    out += buf;
    if (!std::strpbrk(buf, ".eEn")) out += ".0";
    break;
  }
  case String: dump_string(out, s_); break;
  case Binary: {
    // same shape nlohmann uses when dumping a binary value to text
    out += "{\"bytes\":[";
    for (size_t i = 0; i < s_.size(); ++i) {
      if (i) out.push_back(',');
      out += std::to_string((unsigned)(unsigned char)s_[i]);
    }
    out += "],\"subtype\":null}";
    break;
  }
  case Array:
    if (arr_.empty()) { out += "[]"; break; }
    out.push_back('[');
    for (size_t i = 0; i < arr_.size(); ++i) {
      if (i) out.push_back(',');
      nl(depth + 1);
      arr_[i].dump_to(out, indent, depth + 1);
    }
    nl(depth);
    out.push_back(']');
    break;
  case Object: {
    if (obj_.empty()) { out += "{}"; break; }
    out.push_back('{');
    bool first = true;
    for (const auto& kv : obj_) {
      if (!first) out.push_back(',');
      first = false;
      nl(depth + 1);
      dump_string(out, kv.first);
      out += indent >= 0 ? ": " : ":";
      kv.second.dump_to(out, indent, depth + 1);
    }
    nl(depth);
    out.push_back('}');
    break;
  }
  }
}

std::string Json::dump(int indent) const
{
  std::string out;
  dump_to(out, indent, 0);
  return out;
}

// ------------------------------------------------------------------------------------------ BSON
namespace {
void put_i32(std::vector<uint8_t>& o, int32_t v) { for (int i = 0; i < 4; ++i) o.push_back((uint8_t)((uint32_t)v >> (8 * i))); }
void put_i64(std::vector<uint8_t>& o, int64_t v) { for (int i = 0; i < 8; ++i) o.push_back((uint8_t)((uint64_t)v >> (8 * i))); }
void patch_i32(std::vector<uint8_t>& o, size_t at, int32_t v) { for (int i = 0; i < 4; ++i) o[at + i] = (uint8_t)((uint32_t)v >> (8 * i)); }

constexpr int kBsonMaxDepth = 64;   // params.json nests 3 deep; a corrupt or crafted file must not recurse the stack away

struct BsonReader {
  const uint8_t* p;
  const uint8_t* end;   // end of the enclosing document while its elements are read (of the buffer at the top)
  int depth = 0;
  [[noreturn]] void fail(const char* m) const { throw std::runtime_error(std::string("bson parse error: ") + m); }
  void need(size_t n) const { if ((size_t)(end - p) < n) fail("unexpected end of input"); }
  int32_t i32() { need(4); uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)p[i] << (8 * i); p += 4; return (int32_t)v; }
  int64_t i64() { need(8); uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i); p += 8; return (int64_t)v; }
  std::string cstr()
  {
    const uint8_t* b = p;
    while (p < end && *p) ++p;
    if (p >= end) fail("unterminated key");
    std::string s((const char*)b, (const char*)p);
    ++p;
    return s;
  }
  Json document(bool as_array)
  {
    if (++depth > kBsonMaxDepth) fail("documents nested too deeply");
    const uint8_t* start = p;
    const int32_t len = i32();
    if (len < 5 || (size_t)len > (size_t)(end - start)) fail("bad document length");   // `end` is the parent's end: a child cannot overrun it
    const uint8_t* doc_end = start + len;
    const uint8_t* parent_end = end;
    end = doc_end;   // strings, binaries and nested documents of this document are bounded by it
    Json out = as_array ? Json::array() : Json::object();
    while (p < doc_end - 1) {
      need(1);
      const uint8_t t = *p++;
      const std::string key = cstr();
      Json v;
      switch (t) {
      case 0x01: { const int64_t raw = i64(); double d; std::memcpy(&d, &raw, 8); v = Json(d); break; }
      case 0x02: {
        const int32_t n = i32();
        if (n < 1) fail("bad string length");
        need((size_t)n);
        v = Json(std::string((const char*)p, (size_t)n - 1));
        p += n;
        break;
      }
      case 0x03: v = document(false); break;
      case 0x04: v = document(true); break;
      case 0x05: {
        const int32_t n = i32();
        if (n < 0) fail("bad binary length");
        need((size_t)n + 1);
        ++p;  // subtype
        v = Json::binary(p, (size_t)n);
        p += n;
        break;
      }
      case 0x08: need(1); v = Json(*p++ != 0); break;
      case 0x0A: v = Json(); break;
      case 0x10: v = Json((int64_t)i32()); break;
      case 0x11:
      case 0x12: v = Json(i64()); break;
      default: fail("unsupported element type");
      }
      if (as_array) out.push_back(v);
      else out[key] = v;
    }
    if (p != doc_end - 1 || *p != 0) fail("bad document terminator");
    ++p;
    end = parent_end;
    --depth;
    return out;
  }
};
}  // namespace

void Json::bson_element(std::vector<uint8_t>& o, const std::string& key) const
{
  // a BSON key is a C string (nlohmann::json: out_of_range.409 "BSON key cannot contain code point U+0000")
  if (key.find('\0') != std::string::npos) throw std::runtime_error("bson: a key cannot contain the code point U+0000");
  auto header = [&](uint8_t t) {
    o.push_back(t);
    o.insert(o.end(), key.begin(), key.end());
    o.push_back(0);
  };
  switch (type_) {
  case Null: header(0x0A); break;
  case Bool: header(0x08); o.push_back(b_ ? 1 : 0); break;
  case Int:
    if (i_ >= INT32_MIN && i_ <= INT32_MAX) { header(0x10); put_i32(o, (int32_t)i_); }
    else { header(0x12); put_i64(o, i_); }
    break;
  case Double: { header(0x01); int64_t raw; std::memcpy(&raw, &d_, 8); put_i64(o, raw); break; }
  case String:
    if (s_.size() >= (size_t)INT32_MAX) throw std::runtime_error("bson: string of 2 GiB or more cannot be encoded");
    header(0x02);
    put_i32(o, (int32_t)s_.size() + 1);
    o.insert(o.end(), s_.begin(), s_.end());
    o.push_back(0);
    break;
  case Binary:
    // a BSON length is an int32 (nlohmann throws out_of_range here): a parameter blob of 2 GiB or more has no params.json form
    if (s_.size() > (size_t)INT32_MAX) throw std::runtime_error("bson: binary of more than 2 GiB - 1 bytes cannot be encoded (" + std::to_string(s_.size()) + " bytes)");
    header(0x05);
    put_i32(o, (int32_t)s_.size());
    o.push_back(0x00);
    o.insert(o.end(), s_.begin(), s_.end());
    break;
  case Array:
  case Object: header(type_ == Array ? 0x04 : 0x03); bson_document(o); break;
  }
}

void Json::bson_document(std::vector<uint8_t>& o) const
{
  const size_t at = o.size();
  put_i32(o, 0);
  if (type_ == Array) {
    for (size_t i = 0; i < arr_.size(); ++i) arr_[i].bson_element(o, std::to_string(i));
  } else {
    for (const auto& kv : obj_) kv.second.bson_element(o, kv.first);
  }
  o.push_back(0);
  if (o.size() - at > (size_t)INT32_MAX) throw std::runtime_error("bson: document of more than 2 GiB - 1 bytes cannot be encoded");
  patch_i32(o, at, (int32_t)(o.size() - at));
}

std::vector<uint8_t> Json::to_bson() const
{
  if (type_ != Object) throw std::runtime_error("bson: top-level value must be an object");
  std::vector<uint8_t> o;
  bson_document(o);
  return o;
}

Json Json::from_bson(const uint8_t* data, size_t size)
{
  BsonReader r{data, data + size, 0};
  return r.document(false);
}

static std::string read_file(const std::string& filename)
{
  std::ifstream f(filename, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open file: " + filename);
  return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

Json Json::load_text_file(const std::string& filename)
{
  const std::string s = read_file(filename);
  return parse_text(s.data(), s.size());
}

Json Json::load_bson_file(const std::string& filename)
{
  const std::string s = read_file(filename);
  return from_bson((const uint8_t*)s.data(), s.size());
}

}  // namespace vnr
