// tests/test_json_fuzz.py builds this with csrc/json.cpp under -fsanitize=address,undefined and feeds it the corpus the Python
// worker recorded: every input through the reader of its format, and whatever parses through both writers and back.
// A heap overrun, a use after free or undefined arithmetic in the parser aborts the process; the test sees the exit code.
#include "json.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  size_t at = 0, n_ok = 0, n_err = 0;
  while (at + 5 <= all.size()) {
    const uint8_t format = (uint8_t)all[at];
    uint32_t len;
    std::memcpy(&len, &all[at + 1], 4);
    at += 5;
    if (at + len > all.size()) return 3;
    // a copy of exactly `len` bytes on the heap: one byte beyond it is a sanitizer report
    std::vector<uint8_t> in(all.begin() + (long)at, all.begin() + (long)(at + len));
    at += len;
    try {
      vnr::Json v = format == 0 ? vnr::Json::parse_text((const char*)in.data(), in.size()) : vnr::Json::from_bson(in.data(), in.size());
      const std::string text = v.dump(2);
      vnr::Json again = vnr::Json::parse_text(text);
      if (v.is_object()) {
        const std::vector<uint8_t> b = v.to_bson();
        vnr::Json back = vnr::Json::from_bson(b.data(), b.size());
        if (back.dump() != again.dump()) { std::fprintf(stderr, "round trip differs\n"); return 4; }
      }
      ++n_ok;
    } catch (const std::exception&) {
      ++n_err;
    }
  }
  std::printf("%zu parsed, %zu refused\n", n_ok, n_err);
  return 0;
}
