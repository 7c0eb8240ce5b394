// Image decode/encode for textures and render output.
//
// The reference decodes textures with stb_image 2.16's stbi_loadf (reference: src/world_load.cpp:131-157,
// external/include/stb/stb_image.h) and saves PNGs through Qt (reference: qt/mainwindow.cpp:93-99). Neither
// library is vendored here. These decoders are written from scratch but follow that decoder's ARITHMETIC, so the
// texels are the reference's bit for bit (byte work: tolerance 0) — checked against the reference's own header
// compiled unchanged (oracle/_ref, tests/test_oracle_pins.py) and against its frozen outputs
// (tests/golden/ref_pins.json "images") on every shipped texture and 148 synthetic files:
//   JPEG  baseline / extended / progressive Huffman, 1, 3 or 4 components (YCbCr, RGB, CMYK, YCCK), sampling
//         factors 1-4, restart intervals; integer IDCT with 12-bit constants, the triangle-filter chroma
//         upsamplers, fixed-point YCbCr -> RGB
//   PNG   every colour type x bit depth, Adam7, tRNS, palettes; 16-bit samples keep their high byte
//   TGA   true-colour / grey / colour-mapped, raw or RLE, 8/15/16/24/32 bits, either origin
// BMP / GIF / PSD / PIC / PNM / HDR files (which stb would also read) are recognised and refused by name.
// loadImageFloat() applies stbi_loadf's LDR->HDR rule (stb_image.h:1565-1580) as the reference's C++ build
// evaluates it: colour channels (float)(pow(v/255.0f, 2.2f) * 1.0f) with pow's float overload, alpha v/255.0f
// (alpha = last channel when the channel count is even).
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
