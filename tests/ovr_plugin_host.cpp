// Test host of the OVR plugin's library-facing half (ovr_plugin/device_nnvolume_amd.h): plays the part of OVR's main loop on plain data.
// usage: ovr_plugin_host <volume.raw (uint16, x fastest)> <nx> <ny> <nz> <out.raw> ; writes two frames (width x height vec4f each): the
// first after init + commit + 3 accumulated render calls, the second after a transfer-function and sampling-rate change (commit) and one render.
// Exit codes: 0 ok, 42 no GPU (std::runtime_error from the first call that needs one), 1 anything else.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "device_nnvolume_amd.h"

int main(int argc, char** argv)
{
  if (argc < 6) return 1;
  const int nx = std::atoi(argv[2]), ny = std::atoi(argv[3]), nz = std::atoi(argv[4]);
  std::vector<uint16_t> voxels((size_t)nx * ny * nz);
  { std::ifstream f(argv[1], std::ios::binary); if (!f.read((char*)voxels.data(), (std::streamsize)(voxels.size() * 2))) return 1; }
  try {
    vnr_amd_plugin::NNVolumeDevice dev;
    vnr_amd_plugin::StructuredVolume v;
    v.data = voxels.data(); v.dims[0] = nx; v.dims[1] = ny; v.dims[2] = nz; v.value_type = VNR_AMD_TYPE_UINT16;
    v.grid_origin[0] = -0.5f * nx; v.grid_origin[1] = -0.5f * ny; v.grid_origin[2] = -0.5f * nz;   // the volume centred at the origin, unit spacing
    const float colors[] = {0, 0, 1, 0, 1, 0, 1, 1, 0, 1, 0, 0};
    const float opac[] = {0.0f, 0.02f, 0.1f, 0.3f, 0.6f};
    dev.init(v, colors, 4, opac, 5, 0.0f, 65535.0f);
    const int W = 160, H = 120;
    dev.resize(W, H);
    const float from[3] = {1.6f * nx, 1.1f * ny, -1.9f * nz}, at[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    dev.set_camera(from, at, up);
    size_t bytes = 0;
    const float* d_pixels = nullptr;
    for (int k = 0; k < 3; ++k) { dev.render(); d_pixels = dev.mapframe(&bytes); }
    std::vector<float> frame(bytes / sizeof(float));
    if (vnrAmdMemcpyD2H(frame.data(), d_pixels, bytes) != VNR_AMD_OK) return 1;
    std::ofstream out(argv[5], std::ios::binary);
    out.write((const char*)frame.data(), (std::streamsize)bytes);
    // a commit with a new transfer function (colors + (position, alpha) pairs, as params.tfn delivers them) and sampling rate
    const float colors2[] = {1, 1, 1, 1, 0.5f, 0};
    const float alphas2[] = {0.0f, 0.0f, 0.5f, 0.05f, 1.0f, 0.8f};
    dev.set_transfer_function(colors2, 2, alphas2, 3, 8000.0f, 60000.0f);
    dev.set_volume_sampling_rate(2.0f);
    dev.render();
    d_pixels = dev.mapframe(&bytes);
    if (vnrAmdMemcpyD2H(frame.data(), d_pixels, bytes) != VNR_AMD_OK) return 1;
    out.write((const char*)frame.data(), (std::streamsize)bytes);
    return 0;
  } catch (const std::runtime_error& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return std::string(e.what()).find("HIP") != std::string::npos || std::string(e.what()).find("device") != std::string::npos ? 42 : 1;
  }
}
