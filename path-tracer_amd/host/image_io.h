// Image decode/encode for textures and render output.
//
// The reference decodes textures with stb_image 2.16's stbi_loadf (reference:
// src/world_load.cpp:131-157) and saves PNGs through Qt (reference:
// qt/mainwindow.cpp:93-99). Neither library is vendored here; these are fresh
// decoders for the three formats the reference's assets use:
//   PNG  (zlib inflate; 8/16-bit gray, gray+alpha, RGB, RGBA, palette; non-interlaced)
//   TGA  (types 2/3/10/11, 8/24/32 bit, either origin)
//   JPEG (baseline / extended-sequential Huffman, 1 or 3 components, h/v sampling 1-2)
// loadImageFloat() applies stbi_loadf's LDR->HDR rule (reference:
// external/include/stb/stb_image.h:1565-1580): colour channels
// (float)pow(v/255.0f, 2.2f), alpha v/255.0f (alpha = last channel when the channel
// count is even).
#pragma once
#include <string>
#include <vector>

// 8-bit decode, top-left origin, `channels` = the file's own channel count (1..4).
bool loadImage8(const std::string& path, int& width, int& height, int& channels, std::vector<unsigned char>& pixels,
                std::string& err);
// stbi_loadf equivalent (req_comp = 0).
bool loadImageFloat(const std::string& path, int& width, int& height, int& channels, std::vector<float>& pixels,
                    std::string& err);

// 8-bit RGB(A) PNG writer (zlib), rows top to bottom. comp = 3 or 4.
bool writePNG(const std::string& path, int width, int height, int comp, const unsigned char* pixels, std::string& err);
// little-endian colour PFM ("PF"), rows written bottom to top as the format requires.
bool writePFM(const std::string& path, int width, int height, const float* rgb, std::string& err);
