// Host-side image textures: see yafaray_image.cpp
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace yafimg {

enum { kSrgb = 0, kXyz = 1, kLinearRgb = 2, kRawManualGamma = 3 };          // ColorSpace, include/common/color.h
enum { kOptNone = 0, kOptOptimized = 1, kOptCompressed = 2 };                 // TextureOptimization

struct Image
{
	// in: how the file's colours are to be read and kept (ImageHandler::setColorSpace / setTextureOptimization / setGrayScaleSetting)
	int color_space = kRawManualGamma; float gamma = 1.f; int optimization = kOptOptimized; bool grayscale = false;
	// out
	int width = 0, height = 0, channels = 0; bool has_alpha = false;
	std::vector<float> texels;         // height * width * 4: what ImageHandler::getPixel(x, y) returns, row y = the handler's row y
};

// decodes `path` by its extension (tga, hdr, png); false + err for anything else or a damaged file
bool load(const std::string &path, Image &img, std::string &err);

} // namespace yafimg
