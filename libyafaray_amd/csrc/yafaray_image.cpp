// Host-side image textures: file decoders and the reference's image-buffer storage semantics.
//
// The device samples float RGBA texels that hold exactly what the reference's ImageHandler::getPixel would return:
// a decoder reads the file's pixels as the reference's handler does (same order, same integer -> float conversion),
// linearises them from the texture's colour space (ImageBuffer::setColor(x, y, col, color_space, gamma),
// include/imagehandler/imagehandler.h:186-198; Rgb::linearRgbFromColorSpace, include/common/color.h:366-386) and pushes
// them through the buffer's storage format ("optimized": 10 bits per colour channel, 8 bits of alpha;
// include/utility/util_image_buffers.h:186-248; "compressed": 7-7-7-3 / 5-6-5; "none": float).
//
//   TGA  src/imagehandler/imagehandler_tga.cc (true colour 15/16/24/32 bit and grey 8/16 bit, raw or RLE; colour-mapped
//        files are refused)
//   HDR  src/imagehandler/imagehandler_hdr.cc (Radiance RGBE, flat / old RLE / adaptive RLE scanlines)
//   PNG  src/imagehandler/imagehandler_png.cc reads through libpng; here the container is parsed directly and the
//        pixel stream inflated with zlib (8 / 16 bit grey, grey + alpha, RGB, RGBA, palette; non-interlaced)
//   JPEG, TIFF, OpenEXR: no decoder in this build -> the texture is refused with a diagnostic.
#include "yafaray_image.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <sstream>
#include <stdexcept>

#include <zlib.h>

namespace yafimg {

namespace {

// fPow__ (util_math_optimizations.h:116-142,176-183; FAST_MATH is on in the reference's build)
float f_exp2(float x)
{
	x = std::min(x, 129.00000f);
	x = std::max(x, -126.99999f);
	const int ipart = (int)(x - 0.5f);
	const float p = (x - (float)ipart);
	const int bits = (int)((unsigned)(ipart + 127) << 23);
	float expi; std::memcpy(&expi, &bits, 4);
	const float poly = (p * (p * (p * (p * (p * 1.8775767e-3f + 8.9893397e-3f) + 5.5826318e-2f) + 2.4015361e-1f) + 6.9315308e-1f) + 9.9999994e-1f);
	return expi * poly;
}
float f_log2(float x)
{
	int i; std::memcpy(&i, &x, 4);
	const float e = (float)(((i & 0x7F800000) >> 23) - 127);
	const int mi = (i & 0x7FFFFF) | 0x3F800000;
	float m; std::memcpy(&m, &mi, 4);
	const float a = m * -3.4436006e-2f + 3.1821337e-1f;
	const float b = m * a + -1.2315303f;
	const double c = (double)(m * b) + 2.5988452;
	const double d = (double)m * c + (double)-3.3241990f;
	const double ee = (double)m * d + (double)3.1157899f;
	return ((float)ee * (m - 1.0f) + e);
}
float f_pow(float a, float b) { return f_exp2(f_log2(a) * b); }

// Rgb::linearRgbFromColorSpace, color.h:366-386 (alpha is left alone)
void linearise(float c[4], int color_space, float gamma)
{
	if(color_space == kSrgb)
	{
		for(int k = 0; k < 3; ++k) c[k] = (c[k] <= 0.04045f) ? (c[k] / 12.92f) : f_pow(((c[k] + 0.055f) / 1.055f), 2.4f);
	}
	else if(color_space == kXyz)
	{
		static const float m[3][3] = {{3.2406255f, -1.537208f, -0.4986286f}, {-0.9689307f, 1.8757561f, 0.0415175f}, {0.0557101f, -0.2040211f, 1.0569959f}};
		const float o[3] = {c[0], c[1], c[2]};
		for(int k = 0; k < 3; ++k) c[k] = m[k][0] * o[0] + m[k][1] * o[1] + m[k][2] * o[2];
	}
	else if(color_space == kRawManualGamma && gamma != 1.f)
		for(int k = 0; k < 3; ++k) c[k] = f_pow(c[k], gamma);
}

// ImageBuffer::setColor followed by ImageBuffer::getColor for a buffer of `channels` channels under `optimization`
void store(const Image &img, const float in[4], float out[4])
{
	const int ch = img.channels;
	if(img.optimization == kOptNone)
	{
		if(ch == 4) { for(int k = 0; k < 4; ++k) out[k] = in[k]; }
		else if(ch == 3) { out[0] = in[0]; out[1] = in[1]; out[2] = in[2]; out[3] = 1.f; }       // Rgb2DImage_t -> Rgba(rgb): alpha 1
		else { const float g = (in[0] + in[1] + in[2]) / 3.f; out[0] = out[1] = out[2] = g; out[3] = 1.f; }
		return;
	}
	if(ch == 1)
	{	// Gray8, util_image_buffers.h:140-162
		const float g = (in[0] + in[1] + in[2]) / 3.f;
		const uint8_t v = (uint8_t)roundf(g * 255.f);
		out[0] = out[1] = out[2] = (float)v / 255.f; out[3] = 1.f;
		return;
	}
	if(img.optimization == kOptOptimized)
	{	// Rgb101010 / Rgba1010108, :186-248: 10 bits per colour, the two high bits kept apart — the value survives
		for(int k = 0; k < 3; ++k) { const uint16_t v = (uint16_t)roundf(in[k] * 1023.f); out[k] = (float)(uint16_t)(v & 0x03FF) / 1023.f; }
		if(ch == 4) { const uint8_t a = (uint8_t)roundf(in[3] * 255.f); out[3] = (float)a / 255.f; }
		else out[3] = 1.f;
		return;
	}
	// compressed
	if(ch == 4)
	{	// Rgba7773, :86-115
		const uint8_t r = (uint8_t)roundf(in[0] * 255.f), g = (uint8_t)roundf(in[1] * 255.f), b = (uint8_t)roundf(in[2] * 255.f), a = (uint8_t)roundf(in[3] * 255.f);
		out[0] = (float)(r & 0xFE) / 254.f; out[1] = (float)(g & 0xFE) / 254.f; out[2] = (float)(b & 0xFE) / 254.f;
		out[3] = (float)(uint8_t)(a & 0xE0) / 224.f;
	}
	else
	{	// Rgb565, :164-184
		const uint8_t r = (uint8_t)roundf(in[0] * 255.f), g = (uint8_t)roundf(in[1] * 255.f), b = (uint8_t)roundf(in[2] * 255.f);
		out[0] = (float)(r & 0xF8) / 248.f; out[1] = (float)(g & 0xFC) / 252.f; out[2] = (float)(b & 0xF8) / 248.f; out[3] = 1.f;
	}
}

void put(Image &img, int x, int y, const float col[4])
{
	if(x < 0 || y < 0 || x >= img.width || y >= img.height) return;
	float c[4] = {col[0], col[1], col[2], col[3]};
	if(!(img.color_space == kLinearRgb || (img.color_space == kRawManualGamma && img.gamma == 1.f))) linearise(c, img.color_space, img.gamma);
	store(img, c, &img.texels[4 * ((size_t)y * (size_t)img.width + (size_t)x)]);
}

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err)
{
	FILE *fp = std::fopen(path.c_str(), "rb");
	if(!fp) { err = "cannot open file " + path; return false; }
	std::fseek(fp, 0, SEEK_END);
	const long n = std::ftell(fp);
	std::fseek(fp, 0, SEEK_SET);
	out.resize((size_t)std::max(n, 0L));
	const size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), fp);
	std::fclose(fp);
	if(got != out.size()) { err = "short read of " + path; return false; }
	return true;
}

// No decoded image may be larger than this many pixels (1 GiB of float4 texels), and none may claim more pixels than its file
// could hold: headers are checked against the file BEFORE anything is allocated.
constexpr uint64_t kMaxPixels = 1ull << 26;
bool size_ok(uint64_t w, uint64_t h, std::string &err, const char *fmt)
{
	if(w == 0 || h == 0 || w * h > kMaxPixels) { err = std::string(fmt) + ": image size out of range (at most " + std::to_string(kMaxPixels) + " pixels)"; return false; }
	return true;
}

void alloc(Image &img)
{
	// an ImageBuffer starts out zeroed (alpha included; Rgba1010108's a_ member starts at 1/255 but every pixel is set)
	img.texels.assign((size_t)img.width * (size_t)img.height * 4, 0.f);
}

// ---- TGA: TgaHandler::loadFromFile, imagehandler_tga.cc:357-530 ---------------------------------------------------
bool load_tga(const std::vector<uint8_t> &d, Image &img, std::string &err)
{
	if(d.size() < 18) { err = "TGA: file too short"; return false; }
	const int id_length = d[0], color_map_type = d[1], image_type = d[2];
	const int cm_entries = d[5] | (d[6] << 8), cm_bits = d[7];
	const int width = d[12] | (d[13] << 8), height = d[14] | (d[15] << 8), bit_depth = d[16], desc = d[17];
	const int alpha_bits = desc & 0x0F;
	const bool from_top = ((desc & 0x20) >> 5) != 0, from_left = ((desc & 0x10) >> 4) != 0;
	bool is_rle = false, is_gray = false;
	switch(image_type)
	{
		case 0: err = "TGA: file has no image data"; return false;
		case 1: case 9: err = "TGA: colour-mapped files are not supported by this decoder"; return false;
		case 2: break;
		case 3: is_gray = true; break;
		case 10: is_rle = true; break;
		case 11: is_gray = true; is_rle = true; break;
		default: err = "TGA: unknown image type"; return false;
	}
	(void)color_map_type; (void)cm_entries;
	if(is_gray) { if(bit_depth != 8 && bit_depth != 16) { err = "TGA: grey images must be 8 or 16 bits deep"; return false; } if(alpha_bits != 8 && bit_depth == 16) { err = "TGA: invalid alpha depth for a 16-bit grey image"; return false; } }
	else
	{
		if(bit_depth != 15 && bit_depth != 16 && bit_depth != 24 && bit_depth != 32) { err = "TGA: unsupported bit depth"; return false; }
		if(alpha_bits != 1 && bit_depth == 16) { err = "TGA: invalid alpha depth for a 16-bit image"; return false; }
		if(alpha_bits != 8 && bit_depth == 32) { err = "TGA: invalid alpha depth for a 32-bit image"; return false; }
	}
	const bool has_alpha = (alpha_bits != 0 || cm_bits == 32);
	if(!size_ok((uint64_t)width, (uint64_t)height, err, "TGA")) return false;
	const size_t bytes_pp = (size_t)((bit_depth + 7) / 8);
	size_t pos = 18 + (size_t)id_length;
	if(pos > d.size()) { err = "TGA: file too short"; return false; }
	// declared pixels against the file: raw data must be there in full; a run-length packet (1 + bytes_pp bytes) yields at most 128 pixels
	if(!is_rle && pos + (size_t)width * (size_t)height * bytes_pp > d.size()) { err = "TGA: pixel data truncated"; return false; }
	if(is_rle && (uint64_t)width * (uint64_t)height > (uint64_t)(d.size() - pos) * 128ull) { err = "TGA: the file is too short for the size its header declares"; return false; }
	img.width = width; img.height = height;
	int n_channels = 3;
	if(cm_bits == 16 || cm_bits == 32 || bit_depth == 16 || bit_depth == 32) n_channels = 4;
	if(img.grayscale) n_channels = 1;
	img.channels = n_channels; img.has_alpha = has_alpha;
	alloc(img);
	// reading order (:433-455)
	int min_x = 0, max_x = width, step_x = 1, min_y = 0, max_y = height, step_y = 1;
	if(!from_top) { min_y = height - 1; max_y = -1; step_y = -1; }
	if(from_left) { min_x = width - 1; max_x = -1; step_x = -1; }
	auto decode = [&](const uint8_t *p, float c[4]) {
		const double inv_255 = 0.00392156862745098039, inv_31 = 0.03225806451612903226;
		if(is_gray && bit_depth == 8) { const float g = (float)(p[0] * inv_255); c[0] = c[1] = c[2] = g; c[3] = g; }     // Rgba(float): all four
		else if(is_gray)
		{
			const unsigned w16 = p[0] | (p[1] << 8);
			const float g = (float)((w16 & 0x00FF) * inv_255);
			c[0] = c[1] = c[2] = g; c[3] = (float)(((w16 & 0xFF00) >> 8) * inv_255);
		}
		else if(bit_depth == 15 || bit_depth == 16)
		{
			const unsigned w16 = p[0] | (p[1] << 8);
			// processColor15 / 16 (:221-240) shift the masked RED bits (RED_MASK 0x003E) by 11 and the BLUE ones (0xF800) by 1 —
			// the shifts of the other channel's mask.  Restated literally: these are the values the reference's texture holds.
			c[0] = (float)(((w16 & 0x003E) >> 11) * inv_31);
			c[1] = (float)(((w16 & 0x07C0) >> 6) * inv_31);
			c[2] = (float)(((w16 & 0xF800) >> 1) * inv_31);
			c[3] = (bit_depth == 16 && has_alpha) ? (float)(w16 & 0x0001) : 1.f;
		}
		else if(bit_depth == 24) { c[0] = (float)(p[2] * inv_255); c[1] = (float)(p[1] * inv_255); c[2] = (float)(p[0] * inv_255); c[3] = 1.f; }
		else { c[0] = (float)(p[2] * inv_255); c[1] = (float)(p[1] * inv_255); c[2] = (float)(p[0] * inv_255); c[3] = (float)(p[3] * inv_255); }
	};
	float c[4];
	if(!is_rle)
	{
		for(int y = min_y; y != max_y; y += step_y)
			for(int x = min_x; x != max_x; x += step_x) { decode(&d[pos], c); put(img, x, y, c); pos += bytes_pp; }
		return true;
	}
	int x = min_x, y = min_y;
	uint8_t color[4] = {0, 0, 0, 0};
	while(pos < d.size() && y != max_y)
	{
		const uint8_t pack = d[pos++];
		const bool rle_pack = (pack & 0x80) != 0;
		const int rep = (int)(pack & 0x7F) + 1;
		if(rle_pack) { if(pos + bytes_pp > d.size()) break; std::memcpy(color, &d[pos], bytes_pp); pos += bytes_pp; }
		for(int i = 0; i < rep; ++i)
		{
			if(!rle_pack) { if(pos + bytes_pp > d.size()) return true; std::memcpy(color, &d[pos], bytes_pp); pos += bytes_pp; }
			decode(color, c); put(img, x, y, c);
			x += step_x;
			if(x == max_x) { x = min_x; y += step_y; if(y == max_y) break; }
		}
	}
	return true;
}

// ---- Radiance HDR: HdrHandler, imagehandler_hdr.cc:52-385 -----------------------------------------------------------
void rgbe_to_rgba(const uint8_t p[4], float c[4])
{	// RgbePixel::getRgba, imagehandler_util_hdr.h:85-97: f = ldexp(1.0, e - 136), narrowed to float by fLdexp__
	if(p[3]) { const float f = (float)std::ldexp((double)1.0f, (int)p[3] - (int)(128 + 8)); c[0] = f * p[0]; c[1] = f * p[1]; c[2] = f * p[2]; c[3] = 1.0f; }
	else { c[0] = c[1] = c[2] = 0.f; c[3] = 1.0f; }
}
bool load_hdr(const std::vector<uint8_t> &d, Image &img, std::string &err)
{
	size_t pos = 0;
	auto getline = [&](std::string &line) -> bool {
		if(pos >= d.size()) return false;
		line.clear();
		while(pos < d.size()) { const char ch = (char)d[pos++]; line.push_back(ch); if(ch == '\n') break; }
		return true;
	};
	std::string line;
	if(!getline(line) || line.find("#?") == std::string::npos) { err = "HDR: not a Radiance RGBE file"; return false; }
	for(;;)
	{
		if(!getline(line)) { err = "HDR: header truncated"; return false; }
		if(line == "" || line == "\n") break;
		const size_t f = line.find("FORMAT=");
		if(f != std::string::npos && line.substr(f + 7).find("32-bit_rle_rgbe") == std::string::npos) { err = "HDR: XYZE files are not supported, only RGBE"; return false; }
	}
	if(!getline(line)) { err = "HDR: no resolution line"; return false; }
	std::vector<std::string> tok;
	{ std::istringstream is(line); std::string t; while(is >> t) tok.push_back(t); }
	if(tok.size() < 4) { err = "HDR: bad resolution line"; return false; }
	const bool y_first = tok[0].find("Y") != std::string::npos;
	int wi = 3, hi = 1, xi = 2, yi = 0, f = 0, s = 1;
	if(!y_first) { wi = 1; hi = 3; xi = 0; yi = 2; f = 1; s = 0; }
	const int width = std::atoi(tok[(size_t)wi].c_str()), height = std::atoi(tok[(size_t)hi].c_str());
	if(width <= 0 || height <= 0) { err = "HDR: bad image size"; return false; }
	if(!size_ok((uint64_t)width, (uint64_t)height, err, "HDR")) return false;
	// every scanline takes at least 4 bytes of the file, and an adaptive run-length scanline at least 8 bytes per 127 pixels
	{
		const uint64_t rest = (uint64_t)(d.size() - std::min(pos, d.size()));
		const uint64_t lines = (uint64_t)(y_first ? height : width), per_line = (uint64_t)(y_first ? width : height);
		if(lines * std::max<uint64_t>(4, (per_line / 127) * 8) > rest + 8) { err = "HDR: the file is too short for the size its header declares"; return false; }
	}
	const bool from_left = tok[(size_t)xi].find("+") != std::string::npos, from_top = tok[(size_t)yi].find("-") != std::string::npos;
	int mn[2], mx[2], st[2];
	mn[f] = 0; mx[f] = height; st[f] = 1;
	mn[s] = 0; mx[s] = width; st[s] = 1;
	if(!from_left) { mn[s] = width - 1; mx[s] = -1; st[s] = -1; }
	if(!from_top) { mn[f] = height - 1; mx[f] = -1; st[f] = -1; }
	img.width = width; img.height = height; img.has_alpha = false;
	img.channels = img.grayscale ? 1 : 4;           // :72-76: four channels although alpha is never read
	alloc(img);
	const int scan_width = y_first ? width : height;
	std::vector<uint8_t> scan((size_t)scan_width * 4);
	float c[4];
	// old RLE / flat scanline, readOrle :247-305.  The reference's loop that copies the scanline into the image steps by
	// header_.max_[1] instead of header_.step_[1] (:296), so it stores pixel 0 of the scanline only: restated as it is.
	auto read_orle = [&](int y) -> bool {
		int rshift = 0, x = mn[1];
		if(x < 0) x = 0;
		while(x < scan_width)
		{
			if(pos + 4 > d.size()) return false;
			const uint8_t *p = &d[pos]; pos += 4;
			if(p[0] == 1 && p[1] == 1 && p[2] == 1)
			{
				int count = (int)p[3] << rshift;
				if(count > scan_width - x || x == 0) return false;
				while(count--) { std::memcpy(&scan[(size_t)x * 4], &scan[(size_t)(x - 1) * 4], 4); ++x; }
				rshift += 8;
			}
			else { std::memcpy(&scan[(size_t)x * 4], p, 4); ++x; rshift = 0; }
		}
		int j = 0;
		for(int xx = mn[1]; xx != mx[1]; xx += mx[1])
		{
			rgbe_to_rgba(&scan[(size_t)j * 4], c);
			if(y_first) put(img, xx, y, c); else put(img, y, xx, c);
			++j;
			if(mx[1] <= 0) break;          // a zero / negative "step" would not terminate in the reference either
		}
		return true;
	};
	auto read_arle = [&](int y, int sw) -> bool {
		for(int chan = 0; chan < 4; ++chan)
		{
			int j = 0;
			while(j < sw)
			{
				if(pos >= d.size()) return false;
				uint8_t count = d[pos++];
				if(count > 128)
				{
					count &= 0x7F;
					if(count + j > sw || pos >= d.size()) return false;
					const uint8_t col = d[pos++];
					while(count--) scan[(size_t)(j++) * 4 + (size_t)chan] = col;
				}
				else
				{
					if(count + j > sw) return false;
					while(count--) { if(pos >= d.size()) return false; scan[(size_t)(j++) * 4 + (size_t)chan] = d[pos++]; }
				}
			}
		}
		int j = 0;
		for(int xx = mn[1]; xx != mx[1]; xx += st[1])
		{
			rgbe_to_rgba(&scan[(size_t)j * 4], c);
			if(y_first) put(img, xx, y, c); else put(img, y, xx, c);
			++j;
		}
		return true;
	};
	if(scan_width < 8 || scan_width > 0x7fff)
	{
		for(int y = mn[0]; y != mx[0]; y += st[0]) if(!read_orle(y)) { err = "HDR: error reading an uncompressed scanline"; return false; }
		return true;
	}
	for(int y = mn[0]; y != mx[0]; y += st[0])
	{
		if(pos + 4 > d.size()) { err = "HDR: error reading a scanline start"; return false; }
		const uint8_t *p = &d[pos];
		if(p[0] == 2 && p[1] == 2 && (int)((p[2] << 8) | p[3]) < 0x8000)
		{
			const int cnt = (int)((p[2] << 8) | p[3]);
			pos += 4;
			if(cnt > scan_width) { err = "HDR: invalid ARLE scanline width"; return false; }
			if(!read_arle(y, cnt)) { err = "HDR: error reading an ARLE scanline"; return false; }
		}
		else if(!read_orle(y)) { err = "HDR: error reading an RLE scanline"; return false; }
	}
	return true;
}

// ---- PNG: the pixel conversion of PngHandler::fillReadBuffer, imagehandler_png.cc:310-456 ---------------------------
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
bool load_png(const std::vector<uint8_t> &d, Image &img, std::string &err)
{
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
	if(d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) { err = "PNG: bad signature"; return false; }
	size_t pos = 8;
	uint32_t w = 0, h = 0; int bit_depth = 0, color_type = 0, interlace = 0;
	std::vector<uint8_t> idat, plte, trns;
	bool have_ihdr = false;
	while(pos + 12 <= d.size())
	{
		const uint32_t n = be32(&d[pos]);
		const char *type = (const char *)&d[pos + 4];
		if(pos + 12 + (size_t)n > d.size()) { err = "PNG: chunk truncated"; return false; }
		const uint8_t *body = &d[pos + 8];
		if(!std::memcmp(type, "IHDR", 4) && n >= 13) { w = be32(body); h = be32(body + 4); bit_depth = body[8]; color_type = body[9]; interlace = body[12]; have_ihdr = true; }
		else if(!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + n);
		else if(!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + n);
		else if(!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + n);
		else if(!std::memcmp(type, "IEND", 4)) break;
		pos += 12 + (size_t)n;
	}
	if(!have_ihdr || w == 0 || h == 0 || w > 65535 || h > 65535) { err = "PNG: bad header"; return false; }
	if(interlace) { err = "PNG: interlaced files are not supported by this decoder"; return false; }
	int src_chan;     // channels in the file's rows
	switch(color_type)
	{
		case 0: src_chan = 1; break;
		case 2: src_chan = 3; break;
		case 3: src_chan = 1; break;
		case 4: src_chan = 2; break;
		case 6: src_chan = 4; break;
		default: err = "PNG: colour type not supported"; return false;
	}
	if(!(bit_depth == 8 || bit_depth == 16 || ((color_type == 0 || color_type == 3) && (bit_depth == 1 || bit_depth == 2 || bit_depth == 4)))) { err = "PNG: bit depth not supported"; return false; }
	if(color_type == 3 && bit_depth == 16) { err = "PNG: a palette image cannot be 16 bits deep"; return false; }
	if(!size_ok(w, h, err, "PNG")) return false;
	// deflate expands at most ~1032:1: a file whose IDAT cannot inflate to the declared size is refused before anything is allocated
	if(((uint64_t)w * (uint64_t)src_chan * (uint64_t)bit_depth + 7) / 8 * (uint64_t)h > (uint64_t)idat.size() * 1100ull + 1024ull) { err = "PNG: the file is too short for the size its header declares"; return false; }
	const size_t bpp_bits = (size_t)src_chan * (size_t)bit_depth;
	const size_t stride = ((size_t)w * bpp_bits + 7) / 8, bpp = std::max<size_t>(1, bpp_bits / 8);
	std::vector<uint8_t> raw((stride + 1) * (size_t)h);
	{
		uLongf out_len = (uLongf)raw.size();
		const int zr = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
		if(zr != Z_OK || out_len != raw.size()) { err = "PNG: pixel data does not inflate to the image's size"; return false; }
	}
	// undo the row filters
	std::vector<uint8_t> pix(stride * (size_t)h);
	for(uint32_t y = 0; y < h; ++y)
	{
		const uint8_t ft = raw[(stride + 1) * y];
		const uint8_t *in = &raw[(stride + 1) * y + 1];
		uint8_t *out = &pix[stride * y];
		const uint8_t *up = y ? &pix[stride * (y - 1)] : nullptr;
		for(size_t i = 0; i < stride; ++i)
		{
			const int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
			int v = in[i];
			switch(ft)
			{
				case 0: break;
				case 1: v += a; break;
				case 2: v += b; break;
				case 3: v += (a + b) >> 1; break;
				case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
				default: err = "PNG: unknown row filter"; return false;
			}
			out[i] = (uint8_t)v;
		}
	}
	// what libpng hands the reference after its transformations (:321-343): palette -> RGB (+ alpha with tRNS); grey below 8 bits
	// -> 8-bit RGB (png_set_gray_to_rgb expands and triples); everything else as stored
	int num_chan = src_chan; int depth = bit_depth; bool has_alpha = color_type == 6;
	std::vector<uint8_t> rows;          // num_chan * (depth / 8) bytes per pixel
	if(color_type == 3)
	{
		num_chan = trns.empty() ? 3 : 4; depth = 8;
		rows.resize((size_t)w * h * (size_t)num_chan);
		for(uint32_t y = 0; y < h; ++y)
			for(uint32_t x = 0; x < w; ++x)
			{
				unsigned idx;
				if(bit_depth == 8) idx = pix[stride * y + x];
				else { const unsigned per = 8u / (unsigned)bit_depth, byte = pix[stride * y + x / per], sh = (per - 1 - x % per) * (unsigned)bit_depth; idx = (byte >> sh) & ((1u << bit_depth) - 1u); }
				uint8_t *o = &rows[((size_t)y * w + x) * (size_t)num_chan];
				for(int k = 0; k < 3; ++k) o[k] = (3 * idx + (unsigned)k < plte.size()) ? plte[3 * idx + (unsigned)k] : 0;
				if(num_chan == 4) o[3] = idx < trns.size() ? trns[idx] : 255;
			}
	}
	else if(color_type == 0 && bit_depth < 8)
	{
		num_chan = 3; depth = 8;
		rows.resize((size_t)w * h * 3);
		const unsigned per = 8u / (unsigned)bit_depth, maxv = (1u << bit_depth) - 1u;
		for(uint32_t y = 0; y < h; ++y)
			for(uint32_t x = 0; x < w; ++x)
			{
				const unsigned byte = pix[stride * y + x / per], sh = (per - 1 - x % per) * (unsigned)bit_depth;
				const uint8_t g = (uint8_t)((((byte >> sh) & maxv) * 255u) / maxv);
				uint8_t *o = &rows[((size_t)y * w + x) * 3];
				o[0] = o[1] = o[2] = g;
			}
	}
	else rows.swap(pix);
	img.width = (int)w; img.height = (int)h; img.has_alpha = has_alpha;
	int n_channels = num_chan;
	if(img.grayscale) n_channels = 1; else if(has_alpha) n_channels = 4;
	// ImageBuffer only knows 1, 3 and 4 channels (imagehandler.cc:33-55): a 2-channel grey + alpha file gets a buffer with no
	// storage in the reference (every pixel reads back as 0); refuse it here rather than reproduce an empty texture
	if(n_channels == 2) { err = "PNG: grey + alpha files have no image buffer in the reference (2 channels); convert to RGBA"; return false; }
	img.channels = n_channels;
	alloc(img);
	const float divisor = depth == 8 ? (float)0.00392156862745098039 : (float)0.00001525902189669642;       // INV_8 / INV_16 narrowed: `float divisor` (:377-379)
	const size_t px_bytes = (size_t)num_chan * (size_t)(depth / 8);
	float c[4];
	for(uint32_t x = 0; x < w; ++x)
		for(uint32_t y = 0; y < h; ++y)
		{
			const uint8_t *p = &rows[((size_t)y * w + x) * px_bytes];
			float v[4] = {0, 0, 0, 0};
			for(int k = 0; k < num_chan; ++k)
				v[k] = depth == 8 ? (float)p[k] * divisor : (float)(uint16_t)((p[2 * k] << 8) | p[2 * k + 1]) * divisor;
			switch(num_chan)
			{
				case 4: c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = v[3]; break;
				case 3: c[0] = v[0]; c[1] = v[1]; c[2] = v[2]; c[3] = 1.f; break;
				case 2: c[0] = c[1] = c[2] = v[0]; c[3] = v[1]; break;
				default: c[0] = c[1] = c[2] = v[0]; c[3] = 1.f; break;
			}
			put(img, (int)x, (int)y, c);
		}
	return true;
}

std::string lower_ext(const std::string &path)
{
	const size_t dot = path.rfind('.');
	std::string e = dot == std::string::npos ? std::string() : path.substr(dot + 1);
	for(char &ch : e) ch = (char)std::tolower((unsigned char)ch);
	return e;
}

} // namespace

static bool load_checked(const std::string &path, Image &img, std::string &err);
// nothing a damaged file can provoke may leave the library as a C++ exception: this is called from extern "C" entry points
bool load(const std::string &path, Image &img, std::string &err)
{
	try { return load_checked(path, img, err); }
	catch(const std::bad_alloc &) { err = "out of memory while decoding " + path; }
	catch(const std::exception &e) { err = std::string("decoding ") + path + " failed: " + e.what(); }
	img.texels.clear(); img.width = img.height = 0;
	return false;
}
static bool load_checked(const std::string &path, Image &img, std::string &err)
{
	const std::string ext = lower_ext(path);
	std::vector<uint8_t> data;
	if(ext == "tga" || ext == "tpic")
	{
		if(!read_file(path, data, err)) return false;
		return load_tga(data, img, err);
	}
	if(ext == "hdr" || ext == "pic")
	{	// ImageTexture::factory :607-613: HDR files are always linear and never optimized
		img.color_space = kLinearRgb; img.optimization = kOptNone;
		if(!read_file(path, data, err)) return false;
		return load_hdr(data, img, err);
	}
	if(ext == "png")
	{
		if(!read_file(path, data, err)) return false;
		return load_png(data, img, err);
	}
	err = "image format \"" + ext + "\" has no decoder in this build (TGA, HDR and PNG are read; JPEG, TIFF and OpenEXR need libraries this image lacks)";
	return false;
}

} // namespace yafimg
