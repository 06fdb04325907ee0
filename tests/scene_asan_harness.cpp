// tests/test_json_fuzz.py builds this with csrc/scene.cpp and csrc/json.cpp (host side only) under -fsanitize=address,undefined and feeds it
// damaged scene documents: the scene / camera / transfer-function readers (serializer.cpp:137-477 restated) on every one of them.
#include "scene.h"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
using namespace vnr;
int main(int argc, char** argv)
{
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<char> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  size_t at = 0, ok = 0, err = 0;
  while (at + 5 <= all.size()) {
    uint8_t fmt = (uint8_t)all[at]; uint32_t len; std::memcpy(&len, &all[at + 1], 4); at += 5;
    if (at + len > all.size()) return 3;
    std::vector<uint8_t> in(all.begin() + (long)at, all.begin() + (long)(at + len)); at += len;
    try {
      Json v = fmt == 0 ? Json::parse_text((const char*)in.data(), in.size()) : Json::from_bson(in.data(), in.size());
      try { SceneVolume sv = parse_scene_volume(v); (void)sv; ++ok; } catch (const std::exception&) { ++err; }
      try { CameraData c; (void)parse_scene_camera(v, c); } catch (const std::exception&) {}
      try { float lo, hi; (void)parse_scene_tfn_range(v, lo, hi); } catch (const std::exception&) {}
    } catch (const std::exception&) { ++err; }
  }
  std::printf("%zu scenes, %zu refused\n", ok, err);
  return 0;
}
