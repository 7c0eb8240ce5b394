// PNG / TGA / baseline-JPEG decoders and PNG / PFM writers. See image_io.h.
#include "image_io.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

typedef unsigned char u8;

bool readFile(const std::string& path, std::vector<u8>& buf) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf.resize(n > 0 ? (size_t)n : 0);
	size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
	fclose(f);
	return got == buf.size();
}

inline unsigned be32(const u8* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }
inline unsigned be16(const u8* p) { return ((unsigned)p[0] << 8) | p[1]; }

// Byte source with the end-of-file rule the reference's decoder relies on: reads past the end give 0.
struct Bytes {
	const u8* d;
	size_t n, p;
	Bytes(const std::vector<u8>& f) : d(f.data()), n(f.size()), p(0) {}
	bool eof() const { return p >= n; }
	int u8v() { return p < n ? d[p++] : 0; }
	int be16v() { int a = u8v(); return (a << 8) | u8v(); }
	int le16v() { int a = u8v(); return a | (u8v() << 8); }
	unsigned be32v() { unsigned a = (unsigned)be16v(); return (a << 16) | (unsigned)be16v(); }
	void skip(long k) { if (k > 0) p = (p + (size_t)k > n) ? n : p + (size_t)k; }
	bool copy(u8* dst, size_t k) {
		if (p + k > n) return false;
		memcpy(dst, d + p, k);
		p += k;
		return true;
	}
};

// ---------------------------------------------------------------- PNG
// Bit-for-bit the 8-bit image stb_image 2.16 hands to stbi__ldr_to_hdr (reference: external/include/stb/stb_image.h
// :4281-4489 unfilter + bit expansion, :4491-4533 Adam7, :4535-4620 tRNS / palette, :4680-4850 chunk rules,
// :1001-1015 16 -> 8 bits by the high byte).
int paeth(int a, int b, int c) {
	int p = a + b - c;
	int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	if (pa <= pb && pa <= pc) return a;
	if (pb <= pc) return b;
	return c;
}

// One (sub)image of w x h pixels out of the inflated stream: unfilter, expand to one sample per channel.
// Samples come out as 16-bit values (8-bit and sub-byte files fill the low byte).
bool pngPass(const u8*& raw, size_t& rawLeft, unsigned w, unsigned h, int nch, int depth, bool grayScale,
             std::vector<unsigned short>& px, std::string& err) {
	const size_t lineBytes = ((size_t)nch * w * depth + 7) >> 3;
	if (rawLeft < (lineBytes + 1) * h) { err = "PNG: not enough pixels"; return false; }
	const size_t bpp = depth >= 8 ? (size_t)nch * depth / 8 : 1; // filter distance
	std::vector<u8> lines(lineBytes * h);
	for (unsigned y = 0; y < h; ++y) {
		const int ft = *raw++;
		if (ft > 4) { err = "PNG: invalid filter"; return false; }
		u8* cur = &lines[lineBytes * y];
		const u8* up = y ? cur - lineBytes : NULL;
		for (size_t x = 0; x < lineBytes; ++x) {
			const int a = x >= bpp ? cur[x - bpp] : 0;
			const int b = up ? up[x] : 0;
			const int c = (up && x >= bpp) ? up[x - bpp] : 0;
			int v = raw[x];
			if (ft == 1) v += a;
			else if (ft == 2) v += b;
			else if (ft == 3) v += (a + b) >> 1;
			else if (ft == 4) v += paeth(a, b, c);
			cur[x] = (u8)v;
		}
		raw += lineBytes;
	}
	rawLeft -= (lineBytes + 1) * h;
	px.resize((size_t)w * h * nch);
	const unsigned mul = grayScale ? (depth == 1 ? 0xffu : depth == 2 ? 0x55u : depth == 4 ? 0x11u : 1u) : 1u;
	for (unsigned y = 0; y < h; ++y) {
		const u8* row = &lines[lineBytes * y];
		unsigned short* o = &px[(size_t)y * w * nch];
		const size_t ns = (size_t)w * nch;
		if (depth == 8) for (size_t i = 0; i < ns; ++i) o[i] = row[i];
		else if (depth == 16) for (size_t i = 0; i < ns; ++i) o[i] = (unsigned short)((row[2 * i] << 8) | row[2 * i + 1]);
		else
			for (size_t i = 0; i < ns; ++i) {
				const size_t bit = i * depth;
				const unsigned s = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
				o[i] = (unsigned short)(u8)(mul * s);
			}
	}
	return true;
}

bool decodePNG(const std::vector<u8>& f, int& w, int& h, int& comp, std::vector<u8>& out, std::string& err) {
	Bytes s(f);
	s.skip(8); // signature checked by the caller
	unsigned W = 0, H = 0;
	int depth = 0, ctype = 0, interlace = 0, nch = 0;
	bool first = true, pal = false, palAlpha = false, keyed = false, iphone = false;
	u8 palette[1024];
	unsigned palLen = 0;
	unsigned short key[3] = {0, 0, 0};
	std::vector<u8> idat;
	for (;;) {
		const unsigned len = s.be32v();
		const unsigned type = s.be32v();
		auto is = [&](const char* t) { return type == (((unsigned)t[0] << 24) | ((unsigned)t[1] << 16) | ((unsigned)t[2] << 8) | (unsigned)t[3]); };
		if (is("CgBI")) { iphone = true; s.skip(len); }
		else if (is("IHDR")) {
			if (!first) { err = "PNG: multiple IHDR"; return false; }
			first = false;
			if (len != 13) { err = "PNG: bad IHDR len"; return false; }
			W = s.be32v(); H = s.be32v();
			if (W > (1u << 24) || H > (1u << 24)) { err = "PNG: too large"; return false; }
			depth = s.u8v();
			if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) { err = "PNG: 1/2/4/8/16-bit only"; return false; }
			ctype = s.u8v();
			if (ctype > 6 || (ctype == 3 && depth == 16)) { err = "PNG: bad ctype"; return false; }
			if (ctype == 3) pal = true;
			else if (ctype & 1) { err = "PNG: bad ctype"; return false; }
			if (s.u8v()) { err = "PNG: bad comp method"; return false; }
			if (s.u8v()) { err = "PNG: bad filter method"; return false; }
			interlace = s.u8v();
			if (interlace > 1) { err = "PNG: bad interlace method"; return false; }
			if (!W || !H) { err = "PNG: 0-pixel image"; return false; }
			nch = pal ? 1 : ((ctype & 2) ? 3 : 1) + ((ctype & 4) ? 1 : 0);
			if ((1u << 30) / W / (pal ? 4 : nch) < H) { err = "PNG: too large"; return false; }
		} else if (is("PLTE")) {
			if (first) { err = "PNG: first not IHDR"; return false; }
			if (len > 256 * 3 || len % 3) { err = "PNG: invalid PLTE"; return false; }
			palLen = len / 3;
			for (unsigned i = 0; i < palLen; ++i) {
				palette[4 * i] = (u8)s.u8v(); palette[4 * i + 1] = (u8)s.u8v(); palette[4 * i + 2] = (u8)s.u8v();
				palette[4 * i + 3] = 255;
			}
		} else if (is("tRNS")) {
			if (first) { err = "PNG: first not IHDR"; return false; }
			if (!idat.empty()) { err = "PNG: tRNS after IDAT"; return false; }
			if (pal) {
				if (!palLen) { err = "PNG: tRNS before PLTE"; return false; }
				if (len > palLen) { err = "PNG: bad tRNS len"; return false; }
				palAlpha = true;
				for (unsigned i = 0; i < len; ++i) palette[4 * i + 3] = (u8)s.u8v();
			} else {
				if (!(nch & 1)) { err = "PNG: tRNS with alpha"; return false; }
				if (len != (unsigned)nch * 2) { err = "PNG: bad tRNS len"; return false; }
				keyed = true;
				const unsigned mul = depth == 1 ? 0xffu : depth == 2 ? 0x55u : depth == 4 ? 0x11u : 1u;
				for (int k = 0; k < nch; ++k) {
					const int v = s.be16v();
					key[k] = depth == 16 ? (unsigned short)v : (unsigned short)(u8)((u8)(v & 255) * mul);
				}
			}
		} else if (is("IDAT")) {
			if (first) { err = "PNG: first not IHDR"; return false; }
			if (pal && !palLen) { err = "PNG: no PLTE"; return false; }
			if (s.p + len > s.n) { err = "PNG: out of data"; return false; }
			idat.insert(idat.end(), s.d + s.p, s.d + s.p + len);
			s.skip(len);
		} else if (is("IEND")) {
			if (first) { err = "PNG: first not IHDR"; return false; }
			break;
		} else {
			if (first) { err = "PNG: first not IHDR"; return false; }
			if (!(type & (1u << 29))) { err = "PNG: unknown critical chunk"; return false; }
			s.skip(len);
		}
		s.be32v(); // CRC, not checked (neither does the reference's decoder)
		if (s.eof() && !is("IEND")) { err = "PNG: truncated"; return false; }
	}
	if (idat.empty()) { err = "PNG: no IDAT"; return false; }
	if (iphone) { err = "PNG: CgBI (raw deflate) files are not supported"; return false; }
	// inflate everything (extra bytes after the image are tolerated, as in the reference's decoder)
	std::vector<u8> raw;
	{
		z_stream zs;
		memset(&zs, 0, sizeof(zs));
		// zlib header checked by hand, then raw inflate: the Adler-32 trailer is not verified (the reference's
		// decoder does not verify it either, external/include/stb/stb_image.h:4049-4060, 4088-4120)
		if (idat.size() < 2 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 32) || (idat[0] & 15) != 8) { err = "PNG: bad zlib header"; return false; }
		if (inflateInit2(&zs, -15) != Z_OK) { err = "PNG: zlib init failed"; return false; }
		zs.next_in = idat.data() + 2;
		zs.avail_in = (uInt)(idat.size() - 2);
		raw.resize(((size_t)W * depth * nch + 7) / 8 * H + H + 64);
		int zr;
		for (;;) {
			zs.next_out = raw.data() + zs.total_out;
			zs.avail_out = (uInt)(raw.size() - zs.total_out);
			zr = inflate(&zs, Z_NO_FLUSH);
			if (zr == Z_STREAM_END) break;
			if (zr != Z_OK && zr != Z_BUF_ERROR) break;
			if (zs.avail_out == 0) raw.resize(raw.size() * 2);
			else if (zs.avail_in == 0) break;
		}
		raw.resize(zs.total_out);
		inflateEnd(&zs);
		if (zr != Z_STREAM_END) { err = "PNG: inflate failed"; return false; }
	}
	std::vector<unsigned short> img((size_t)W * H * nch);
	const u8* rp = raw.data();
	size_t left = raw.size();
	if (!interlace) {
		if (!pngPass(rp, left, W, H, nch, depth, ctype == 0, img, err)) return false;
	} else {
		static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1};
		static const int dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
		std::vector<unsigned short> sub;
		for (int p = 0; p < 7; ++p) {
			const unsigned pw = (W - x0[p] + dx[p] - 1) / dx[p], ph = (H - y0[p] + dy[p] - 1) / dy[p];
			if (W <= (unsigned)x0[p] || H <= (unsigned)y0[p] || !pw || !ph) continue;
			if (!pngPass(rp, left, pw, ph, nch, depth, ctype == 0, sub, err)) return false;
			for (unsigned j = 0; j < ph; ++j)
				for (unsigned i = 0; i < pw; ++i)
					memcpy(&img[(((size_t)j * dy[p] + y0[p]) * W + (size_t)i * dx[p] + x0[p]) * nch], &sub[((size_t)j * pw + i) * nch],
					       sizeof(unsigned short) * nch);
		}
	}
	w = (int)W; h = (int)H;
	const size_t npix = (size_t)W * H;
	if (pal) {
		comp = palAlpha ? 4 : 3;
		out.resize(npix * comp);
		for (size_t i = 0; i < npix; ++i) memcpy(&out[i * comp], &palette[4 * (img[i] & 255)], (size_t)comp);
		return true;
	}
	comp = nch + (keyed ? 1 : 0);
	out.resize(npix * comp);
	const int sh = depth == 16 ? 8 : 0;
	for (size_t i = 0; i < npix; ++i) {
		const unsigned short* v = &img[i * nch];
		u8* o = &out[i * comp];
		for (int c = 0; c < nch; ++c) o[c] = (u8)(v[c] >> sh);
		if (keyed) {
			bool hit = true;
			for (int c = 0; c < nch; ++c) hit = hit && v[c] == key[c];
			o[nch] = hit ? 0 : 255;
		}
	}
	return true;
}

// ---------------------------------------------------------------- TGA
// stb_image 2.16's acceptance test and loader (reference: external/include/stb/stb_image.h:5317-5346, :5368-5560):
// true-colour, grey and colour-mapped images, raw or run-length coded, 8/15/16/24/32 bits.
int tgaChannels(int bits, bool grey, bool& rgb16) {
	rgb16 = false;
	if (bits == 8) return 1;
	if (bits == 16 && grey) return 2;
	if (bits == 15 || bits == 16) { rgb16 = true; return 3; }
	if (bits == 24 || bits == 32) return bits / 8;
	return 0;
}

bool looksLikeTGA(const std::vector<u8>& f) {
	Bytes s(f);
	s.u8v();
	const int cmap = s.u8v();
	if (cmap > 1) return false;
	int t = s.u8v();
	if (cmap == 1) {
		if (t != 1 && t != 9) return false;
		s.skip(4);
		t = s.u8v();
		if (!(t == 8 || t == 15 || t == 16 || t == 24 || t == 32)) return false;
		s.skip(4);
	} else {
		if (!(t == 2 || t == 3 || t == 10 || t == 11)) return false;
		s.skip(9);
	}
	if (s.le16v() < 1 || s.le16v() < 1) return false;
	t = s.u8v();
	if (cmap == 1 && t != 8 && t != 16) return false;
	return t == 8 || t == 15 || t == 16 || t == 24 || t == 32;
}

void tga555(Bytes& s, u8* o) {
	const unsigned v = (unsigned)s.le16v();
	o[0] = (u8)((((v >> 10) & 31) * 255) / 31);
	o[1] = (u8)((((v >> 5) & 31) * 255) / 31);
	o[2] = (u8)(((v & 31) * 255) / 31);
}

bool decodeTGA(const std::vector<u8>& f, int& w, int& h, int& comp, std::vector<u8>& out, std::string& err) {
	Bytes s(f);
	const int idLen = s.u8v(), indexed = s.u8v();
	int type = s.u8v();
	const int palStart = s.le16v(), palLen = s.le16v(), palBits = s.u8v();
	s.le16v(); s.le16v();
	w = s.le16v(); h = s.le16v();
	const int bits = s.u8v(), desc = s.u8v();
	const bool rle = type >= 8;
	if (rle) type -= 8;
	const bool bottomUp = !((desc >> 5) & 1);
	bool rgb16 = false;
	comp = indexed ? tgaChannels(palBits, false, rgb16) : tgaChannels(bits, type == 3, rgb16);
	if (!comp) { err = "TGA: cannot find out the pixel format"; return false; }
	const size_t npix = (size_t)w * h;
	out.assign(npix * comp, 0);
	s.skip(idLen);
	if (!indexed && !rle && !rgb16) {
		for (int y = 0; y < h; ++y) {
			u8* row = &out[(size_t)(bottomUp ? h - 1 - y : y) * w * comp];
			if (!s.copy(row, (size_t)w * comp)) { // short file: what is there, zeros after it
				const size_t have = s.n - s.p;
				memcpy(row, s.d + s.p, have);
				s.p = s.n;
			}
		}
	} else {
		std::vector<u8> palette;
		if (indexed) {
			s.skip(palStart);
			palette.assign((size_t)palLen * comp, 0);
			if (rgb16) for (int i = 0; i < palLen; ++i) tga555(s, &palette[(size_t)i * comp]);
			else if (!s.copy(palette.data(), palette.size())) { err = "TGA: bad palette"; return false; }
		}
		u8 px[4] = {0, 0, 0, 0};
		int run = 0;
		bool repeat = false;
		for (size_t i = 0; i < npix; ++i) {
			bool fetch = true;
			if (rle) {
				if (run == 0) {
					const int c = s.u8v();
					run = 1 + (c & 127);
					repeat = (c >> 7) != 0;
				} else if (repeat) fetch = false;
			}
			if (fetch) {
				if (indexed) {
					int idx = bits == 8 ? s.u8v() : s.le16v();
					if (idx >= palLen) idx = 0;
					for (int c = 0; c < comp; ++c) px[c] = palLen ? palette[(size_t)idx * comp + c] : 0;
				} else if (rgb16) tga555(s, px);
				else for (int c = 0; c < comp; ++c) px[c] = (u8)s.u8v();
			}
			memcpy(&out[i * comp], px, (size_t)comp);
			--run;
		}
		if (bottomUp) {
			const size_t rowb = (size_t)w * comp;
			for (int y = 0; y * 2 < h; ++y)
				std::swap_ranges(&out[rowb * y], &out[rowb * y] + rowb, &out[rowb * (size_t)(h - 1 - y)]);
		}
	}
	if (comp >= 3 && !rgb16)
		for (size_t i = 0; i < npix; ++i) std::swap(out[i * comp], out[i * comp + 2]); // stored BGR(A)
	return true;
}

// ---------------------------------------------------------------- JPEG
// Baseline, extended-sequential and progressive Huffman JPEG with the arithmetic of stb_image 2.16, so that the
// texels are the reference's byte for byte (reference: external/include/stb/stb_image.h — :1905-1955 block decode
// with 16-bit products, :2114-2212 integer IDCT with 12-bit constants and its two rounding points, :3124-3186 /
// :3305-3314 chroma upsampling filters, :3318-3342 fixed-point YCbCr -> RGB, :3524-3681 output rules,
// :1957-2101 + :2768-2785 progressive scans). Its SSE2 kernels are documented as bit-identical to these scalar
// forms; tests/test_oracle_pins.py checks this decoder against the reference build itself.
const u8 kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
inline int zigzagAt(int k) { return k < 64 ? kZigzag[k] : 63; }

struct JHuff {
	u8 sym[256];
	u8 len[257];
	unsigned limit[18]; // (last code of length k) + 1, left-aligned in 16 bits
	int bias[17];       // symbol index = code + bias[length]
	bool set;
	JHuff() : set(false) {}
	bool build(const int count[16]) {
		int k = 0;
		for (int l = 0; l < 16; ++l)
			for (int j = 0; j < count[l]; ++j) {
				if (k >= 256) return false;
				len[k++] = (u8)(l + 1);
			}
		len[k] = 0;
		int code = 0;
		k = 0;
		for (int l = 1; l <= 16; ++l) {
			bias[l] = k - code;
			while (len[k] == l) { ++k; ++code; }
			if (code - 1 >= (1 << l) && bias[l] != k - code) return false;
			limit[l] = (unsigned)code << (16 - l);
			code <<= 1;
		}
		limit[17] = 0xffffffffu;
		set = true;
		return true;
	}
};

struct JBits {
	Bytes* s;
	unsigned acc; // next bit = bit 31
	int have;
	int marker;   // marker met inside entropy-coded data, 0xff = none
	bool dry;     // after a marker the stream yields zero bits
	JBits() : s(NULL), acc(0), have(0), marker(0xff), dry(false) {}
	void reset() { acc = 0; have = 0; dry = false; marker = 0xff; }
	void fill() {
		do {
			const int b = dry ? 0 : s->u8v();
			if (b == 0xff) {
				int c = s->u8v();
				while (c == 0xff) c = s->u8v();
				if (c != 0) { marker = c; dry = true; return; }
			}
			acc |= (unsigned)b << (24 - have);
			have += 8;
		} while (have <= 24);
	}
	unsigned take(int nb) { // nb in 0..16
		if (have < nb) fill();
		if (nb == 0) return 0;
		const unsigned v = acc >> (32 - nb);
		acc <<= nb;
		have -= nb;
		return v;
	}
	int bit() { return (int)take(1); }
	int extend(int nb) { // receive + extend
		if (nb == 0) return 0;
		const int v = (int)take(nb);
		return v < (1 << (nb - 1)) ? v - (1 << nb) + 1 : v;
	}
	int symbol(const JHuff& h) { // -1: code not in the table
		if (have < 16) fill();
		const unsigned top = acc >> 16;
		int l = 1;
		while (l <= 16 && top >= h.limit[l]) ++l;
		if (l > 16) { have -= 16; return -1; }
		if (l > have) return -1;
		const int idx = (int)(acc >> (32 - l)) + h.bias[l];
		acc <<= l;
		have -= l;
		return h.sym[idx & 255];
	}
};

inline int fix12(float x) { return (int)(x * 4096 + 0.5); }

// The 1-D step shared by both passes; inputs are one column or row, outputs the even part e[0..3] (before the
// rounding constant is added) and the odd part o[0..3], so that out[k] = e[k] + o[k], out[7-k] = e[k] - o[k].
inline void idct1d(const int s[8], int e[4], int o[4]) {
	static const int c0541 = fix12(0.5411961f), cm1847 = fix12(-1.847759065f), c0765 = fix12(0.765366865f),
	                 c1175 = fix12(1.175875602f), c0298 = fix12(0.298631336f), c2053 = fix12(2.053119869f),
	                 c3072 = fix12(3.072711026f), c1501 = fix12(1.501321110f), cm0899 = fix12(-0.899976223f),
	                 cm2562 = fix12(-2.562915447f), cm1961 = fix12(-1.961570560f), cm0390 = fix12(-0.390180644f);
	const int z = (s[2] + s[6]) * c0541;
	const int evenLo = z + s[6] * cm1847, evenHi = z + s[2] * c0765;
	const int sum = (s[0] + s[4]) * 4096, dif = (s[0] - s[4]) * 4096;
	e[0] = sum + evenHi; e[3] = sum - evenHi;
	e[1] = dif + evenLo; e[2] = dif - evenLo;
	const int a73 = s[7] + s[3], a51 = s[5] + s[1], a71 = s[7] + s[1], a53 = s[5] + s[3];
	const int z5 = (a73 + a51) * c1175;
	const int p71 = z5 + a71 * cm0899, p53 = z5 + a53 * cm2562, p73 = a73 * cm1961, p51 = a51 * cm0390;
	o[0] = s[1] * c1501 + (p71 + p51);
	o[1] = s[3] * c3072 + (p53 + p73);
	o[2] = s[5] * c2053 + (p53 + p51);
	o[3] = s[7] * c0298 + (p71 + p73);
}

inline u8 clamp255(int v) { return (u8)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void idctBlock(u8* out, int stride, const short coef[64]) {
	int mid[64];
	for (int x = 0; x < 8; ++x) { // columns, keeping two extra bits
		int s[8], e[4], o[4];
		for (int k = 0; k < 8; ++k) s[k] = coef[8 * k + x];
		idct1d(s, e, o);
		for (int k = 0; k < 4; ++k) {
			mid[8 * k + x] = (e[k] + 512 + o[k]) >> 10;
			mid[8 * (7 - k) + x] = (e[k] + 512 - o[k]) >> 10;
		}
	}
	for (int y = 0; y < 8; ++y) { // rows: 12 + 2 + 3 bits to remove, +128 level shift folded into the rounding
		int e[4], o[4];
		idct1d(&mid[8 * y], e, o);
		u8* r = out + (size_t)y * stride;
		const int round = 65536 + (128 << 17);
		for (int k = 0; k < 4; ++k) {
			r[k] = clamp255((e[k] + round + o[k]) >> 17);
			r[7 - k] = clamp255((e[k] + round - o[k]) >> 17);
		}
	}
}

struct JComp {
	int id, h, v, tq, td, ta, pred;
	int pw, ph;   // effective pixels of the component
	int w2, h2;   // allocated plane (whole interleaved MCUs)
	std::vector<u8> plane;
	std::vector<short> coef; // progressive only: w2/8 x h2/8 blocks
};

struct JFile {
	Bytes s;
	JBits br;
	unsigned short qt[4][64];
	JHuff hdc[4], hac[4];
	JComp c[4];
	int nc, W, H, hmax, vmax, mcux, mcuy;
	bool progressive, jfif;
	int adobe, rgbIds, restart, todo, eobrun;
	int scanN, order[4], ss, se, ah, al;
	std::string err;
	JFile(const std::vector<u8>& f) : s(f), nc(0), W(0), H(0), hmax(1), vmax(1), mcux(0), mcuy(0), progressive(false), jfif(false),
	                                    adobe(-1), rgbIds(0), restart(0), todo(0), eobrun(0), scanN(0), ss(0), se(63), ah(0), al(0) {
		br.s = &s;
		memset(qt, 0, sizeof(qt));
	}
	bool fail(const char* m) { err = std::string("JPEG: ") + m; return false; }

	int nextMarker() {
		if (br.marker != 0xff) { const int m = br.marker; br.marker = 0xff; return m; }
		int x = s.u8v();
		if (x != 0xff) return 0xff;
		while (x == 0xff) x = s.u8v();
		return x;
	}

	bool tables(int m) { // DRI, DQT, DHT, APPn, COM
		if (m == 0xff) return fail("expected marker");
		if (m == 0xDD) {
			if (s.be16v() != 4) return fail("bad DRI len");
			restart = s.be16v();
			return true;
		}
		if (m == 0xDB) {
			int L = s.be16v() - 2;
			while (L > 0) {
				const int q = s.u8v(), wide = q >> 4, t = q & 15;
				if (wide > 1) return fail("bad DQT type");
				if (t > 3) return fail("bad DQT table");
				for (int i = 0; i < 64; ++i) qt[t][kZigzag[i]] = (unsigned short)(wide ? s.be16v() : s.u8v());
				L -= wide ? 129 : 65;
			}
			return L == 0 ? true : fail("bad DQT len");
		}
		if (m == 0xC4) {
			int L = s.be16v() - 2;
			while (L > 0) {
				const int q = s.u8v(), tc = q >> 4, th = q & 15;
				if (tc > 1 || th > 3) return fail("bad DHT header");
				int count[16], total = 0;
				for (int i = 0; i < 16; ++i) { count[i] = s.u8v(); total += count[i]; }
				L -= 17;
				JHuff& hf = tc ? hac[th] : hdc[th];
				if (total > 256 || !hf.build(count)) return fail("bad code lengths");
				for (int i = 0; i < total; ++i) hf.sym[i] = (u8)s.u8v();
				L -= total;
			}
			return L == 0 ? true : fail("bad DHT len");
		}
		if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
			int L = s.be16v();
			if (L < 2) return fail("bad APP / COM len");
			L -= 2;
			if (m == 0xE0 && L >= 5) {
				static const char tag[5] = {'J', 'F', 'I', 'F', 0};
				bool ok = true;
				for (int i = 0; i < 5; ++i) ok = (s.u8v() == tag[i]) && ok;
				L -= 5;
				if (ok) jfif = true;
			} else if (m == 0xEE && L >= 12) {
				static const char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
				bool ok = true;
				for (int i = 0; i < 6; ++i) ok = (s.u8v() == tag[i]) && ok;
				L -= 6;
				if (ok) {
					s.u8v(); s.be16v(); s.be16v();
					adobe = s.u8v();
					L -= 6;
				}
			}
			s.skip(L);
			return true;
		}
		return fail("unknown marker");
	}

	bool frame() {
		const int Lf = s.be16v();
		if (Lf < 11) return fail("bad SOF len");
		if (s.u8v() != 8) return fail("only 8-bit samples");
		H = s.be16v();
		if (!H) return fail("no header height");
		W = s.be16v();
		if (!W) return fail("0 width");
		nc = s.u8v();
		if (nc != 1 && nc != 3 && nc != 4) return fail("bad component count");
		if (Lf != 8 + 3 * nc) return fail("bad SOF len");
		rgbIds = 0;
		for (int i = 0; i < nc; ++i) {
			c[i].id = s.u8v();
			if (nc == 3 && c[i].id == "RGB"[i]) ++rgbIds;
			const int q = s.u8v();
			c[i].h = q >> 4; c[i].v = q & 15;
			if (c[i].h < 1 || c[i].h > 4) return fail("bad H");
			if (c[i].v < 1 || c[i].v > 4) return fail("bad V");
			c[i].tq = s.u8v();
			if (c[i].tq > 3) return fail("bad TQ");
			c[i].td = c[i].ta = 0;
			if (c[i].h > hmax) hmax = c[i].h;
			if (c[i].v > vmax) vmax = c[i].v;
		}
		mcux = (W + 8 * hmax - 1) / (8 * hmax);
		mcuy = (H + 8 * vmax - 1) / (8 * vmax);
		for (int i = 0; i < nc; ++i) {
			c[i].pw = (W * c[i].h + hmax - 1) / hmax;
			c[i].ph = (H * c[i].v + vmax - 1) / vmax;
			c[i].w2 = mcux * c[i].h * 8;
			c[i].h2 = mcuy * c[i].v * 8;
			c[i].plane.assign((size_t)c[i].w2 * c[i].h2, 0);
			if (progressive) c[i].coef.assign((size_t)c[i].w2 * c[i].h2, 0);
		}
		return true;
	}

	bool scanHeader() {
		const int Ls = s.be16v();
		scanN = s.u8v();
		if (scanN < 1 || scanN > 4 || scanN > nc) return fail("bad SOS component count");
		if (Ls != 6 + 2 * scanN) return fail("bad SOS len");
		for (int i = 0; i < scanN; ++i) {
			const int id = s.u8v(), q = s.u8v();
			int which = 0;
			while (which < nc && c[which].id != id) ++which;
			if (which == nc) return fail("scan names an unknown component");
			c[which].td = q >> 4; c[which].ta = q & 15;
			if (c[which].td > 3) return fail("bad DC huff");
			if (c[which].ta > 3) return fail("bad AC huff");
			order[i] = which;
		}
		ss = s.u8v(); se = s.u8v();
		const int a = s.u8v();
		ah = a >> 4; al = a & 15;
		if (progressive) {
			if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return fail("bad SOS");
		} else {
			if (ss != 0 || ah != 0 || al != 0) return fail("bad SOS");
			se = 63;
		}
		return true;
	}

	void restartState() {
		br.reset();
		for (int i = 0; i < 4; ++i) c[i].pred = 0;
		todo = restart ? restart : 0x7fffffff;
		eobrun = 0;
	}

	bool blockSequential(JComp& jc, short d[64]) {
		const JHuff& dc = hdc[jc.td]; const JHuff& ac = hac[jc.ta];
		const unsigned short* q = qt[jc.tq];
		const int t = br.symbol(dc);
		if (t < 0 || t > 16) return fail("bad huffman code");
		memset(d, 0, 64 * sizeof(short));
		jc.pred += br.extend(t);
		d[0] = (short)(jc.pred * q[0]);
		for (int k = 1; k < 64;) {
			const int rs = br.symbol(ac);
			if (rs < 0) return fail("bad huffman code");
			const int r = rs >> 4, sz = rs & 15;
			if (sz == 0) {
				if (rs != 0xf0) break;
				k += 16;
			} else {
				k += r;
				const int z = zigzagAt(k++);
				d[z] = (short)(br.extend(sz) * q[z]);
			}
		}
		return true;
	}

	bool blockProgDC(JComp& jc, short d[64]) {
		if (se != 0) return fail("can't merge dc and ac");
		if (ah == 0) {
			memset(d, 0, 64 * sizeof(short));
			const int t = br.symbol(hdc[jc.td]);
			if (t > 16) return fail("bad huffman code");
			jc.pred += t > 0 ? br.extend(t) : 0;
			d[0] = (short)(jc.pred << al);
		} else if (br.bit()) d[0] = (short)(d[0] + (short)(1 << al));
		return true;
	}

	void refine(short& p, short bitv) {
		if (br.bit() && (p & bitv) == 0) p = (short)(p > 0 ? p + bitv : p - bitv);
	}

	bool blockProgAC(JComp& jc, short d[64]) {
		if (ss == 0) return fail("can't merge dc and ac");
		const JHuff& ac = hac[jc.ta];
		if (ah == 0) {
			if (eobrun) { --eobrun; return true; }
			int k = ss;
			do {
				const int rs = br.symbol(ac);
				if (rs < 0) return fail("bad huffman code");
				const int r = rs >> 4, sz = rs & 15;
				if (sz == 0) {
					if (r < 15) {
						eobrun = 1 << r;
						if (r) eobrun += (int)br.take(r);
						--eobrun;
						break;
					}
					k += 16;
				} else {
					k += r;
					d[zigzagAt(k++)] = (short)(br.extend(sz) << al);
				}
			} while (k <= se);
			return true;
		}
		const short bitv = (short)(1 << al);
		if (eobrun) {
			--eobrun;
			for (int k = ss; k <= se; ++k) {
				short& p = d[kZigzag[k]];
				if (p != 0) refine(p, bitv);
			}
			return true;
		}
		int k = ss;
		do {
			const int rs = br.symbol(ac);
			if (rs < 0) return fail("bad huffman code");
			int r = rs >> 4, sz = rs & 15;
			if (sz == 0) {
				if (r < 15) {
					eobrun = (1 << r) - 1;
					if (r) eobrun += (int)br.take(r);
					r = 64;
				}
			} else {
				if (sz != 1) return fail("bad huffman code");
				sz = br.bit() ? bitv : -bitv;
			}
			while (k <= se) {
				short& p = d[kZigzag[k++]];
				if (p != 0) refine(p, bitv);
				else {
					if (r == 0) { p = (short)sz; break; }
					--r;
				}
			}
		} while (k <= se);
		return true;
	}

	// one MCU done: at the end of a restart interval the next thing in the stream must be RSTn
	bool intervalEnds(bool& stop) {
		stop = false;
		if (--todo > 0) return true;
		if (br.have < 24) br.fill();
		if (!(br.marker >= 0xd0 && br.marker <= 0xd7)) { stop = true; return true; }
		restartState();
		return true;
	}

	bool entropy() {
		restartState();
		for (int i = 0; i < scanN; ++i) {
			const JComp& jc = c[order[i]];
			const bool needDC = !progressive || ss == 0, needAC = !progressive || ss != 0;
			if (needDC && !(progressive && ah != 0) && !hdc[jc.td].set) return fail("missing DC Huffman table");
			if (needAC && !hac[jc.ta].set) return fail("missing AC Huffman table");
		}
		short d[64];
		bool stop;
		if (scanN == 1) {
			JComp& jc = c[order[0]];
			const int bw = (jc.pw + 7) >> 3, bh = (jc.ph + 7) >> 3;
			for (int j = 0; j < bh; ++j)
				for (int i = 0; i < bw; ++i) {
					if (!progressive) {
						if (!blockSequential(jc, d)) return false;
						idctBlock(&jc.plane[(size_t)jc.w2 * j * 8 + i * 8], jc.w2, d);
					} else {
						short* blk = &jc.coef[64 * ((size_t)i + (size_t)j * (jc.w2 / 8))];
						if (!(ss == 0 ? blockProgDC(jc, blk) : blockProgAC(jc, blk))) return false;
					}
					intervalEnds(stop);
					if (stop) return true;
				}
			return true;
		}
		for (int j = 0; j < mcuy; ++j)
			for (int i = 0; i < mcux; ++i) {
				for (int k = 0; k < scanN; ++k) {
					JComp& jc = c[order[k]];
					for (int y = 0; y < jc.v; ++y)
						for (int x = 0; x < jc.h; ++x) {
							const int bx = i * jc.h + x, by = j * jc.v + y;
							if (!progressive) {
								if (!blockSequential(jc, d)) return false;
								idctBlock(&jc.plane[(size_t)jc.w2 * by * 8 + bx * 8], jc.w2, d);
							} else if (!blockProgDC(jc, &jc.coef[64 * ((size_t)bx + (size_t)by * (jc.w2 / 8))])) return false;
						}
				}
				intervalEnds(stop);
				if (stop) return true;
			}
		return true;
	}

	bool decode() {
		if (nextMarker() != 0xD8) return fail("no SOI");
		int m = nextMarker();
		while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {
			if (!tables(m)) return false;
			m = nextMarker();
			while (m == 0xff) {
				if (s.eof()) return fail("no SOF");
				m = nextMarker();
			}
		}
		progressive = m == 0xC2;
		if (!frame()) return false;
		m = nextMarker();
		while (m != 0xD9) {
			if (m == 0xDA) {
				if (!scanHeader() || !entropy()) return false;
				if (br.marker == 0xff) // trailing zeros after the scan data
					while (!s.eof())
						if (s.u8v() == 255) { br.marker = s.u8v(); break; }
			} else if (m == 0xDC) { s.be16v(); s.be16v(); }
			else if (!tables(m)) return false;
			m = nextMarker();
		}
		if (progressive)
			for (int n = 0; n < nc; ++n) {
				JComp& jc = c[n];
				const int bw = (jc.pw + 7) >> 3, bh = (jc.ph + 7) >> 3;
				for (int j = 0; j < bh; ++j)
					for (int i = 0; i < bw; ++i) {
						short* blk = &jc.coef[64 * ((size_t)i + (size_t)j * (jc.w2 / 8))];
						for (int k = 0; k < 64; ++k) blk[k] = (short)(blk[k] * qt[jc.tq][k]);
						idctBlock(&jc.plane[(size_t)jc.w2 * j * 8 + i * 8], jc.w2, blk);
					}
			}
		return true;
	}
};

// chroma upsampling of one output row; `near`/`far` are the two source rows that bracket it
const u8* upsampleRow(u8* out, const u8* nearRow, const u8* farRow, int w, int hs, int vs) {
	if (hs == 1 && vs == 1) return nearRow;
	if (hs == 1 && vs == 2) {
		for (int i = 0; i < w; ++i) out[i] = (u8)((3 * nearRow[i] + farRow[i] + 2) >> 2);
		return out;
	}
	if (hs == 2 && vs == 1) {
		const u8* in = nearRow;
		if (w == 1) { out[0] = out[1] = in[0]; return out; }
		out[0] = in[0];
		out[1] = (u8)((in[0] * 3 + in[1] + 2) >> 2);
		for (int i = 1; i < w - 1; ++i) {
			const int n = 3 * in[i] + 2;
			out[2 * i] = (u8)((n + in[i - 1]) >> 2);
			out[2 * i + 1] = (u8)((n + in[i + 1]) >> 2);
		}
		out[2 * (w - 1)] = (u8)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
		out[2 * (w - 1) + 1] = in[w - 1];
		return out;
	}
	if (hs == 2 && vs == 2) {
		if (w == 1) { out[0] = out[1] = (u8)((3 * nearRow[0] + farRow[0] + 2) >> 2); return out; }
		int cur = 3 * nearRow[0] + farRow[0];
		out[0] = (u8)((cur + 2) >> 2);
		for (int i = 1; i < w; ++i) {
			const int prev = cur;
			cur = 3 * nearRow[i] + farRow[i];
			out[2 * i - 1] = (u8)((3 * prev + cur + 8) >> 4);
			out[2 * i] = (u8)((3 * cur + prev + 8) >> 4);
		}
		out[2 * w - 1] = (u8)((cur + 2) >> 2);
		return out;
	}
	for (int i = 0; i < w; ++i)
		for (int j = 0; j < hs; ++j) out[i * hs + j] = nearRow[i];
	return out;
}

inline int ycc(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

void yccToRgbRow(u8* out, const u8* y, const u8* cb, const u8* cr, int count, int step) {
	static const int kRcr = ycc(1.40200f), kGcr = -ycc(0.71414f), kGcb = -ycc(0.34414f), kBcb = ycc(1.77200f);
	for (int i = 0; i < count; ++i, out += step) {
		const int base = (y[i] << 20) + (1 << 19);
		const int r = cr[i] - 128, b = cb[i] - 128;
		const int R = (base + r * kRcr) >> 20;
		const int G = (base + r * kGcr + (int)((unsigned)(b * kGcb) & 0xffff0000u)) >> 20;
		const int B = (base + b * kBcb) >> 20;
		out[0] = clamp255(R); out[1] = clamp255(G); out[2] = clamp255(B);
	}
}

inline u8 mul255(int x, int y) {
	const unsigned t = (unsigned)(x * y + 128);
	return (u8)((t + (t >> 8)) >> 8);
}

bool decodeJPEG(const std::vector<u8>& f, int& w, int& h, int& comp, std::vector<u8>& out, std::string& err) {
	JFile* jp = new JFile(f);
	struct Guard { JFile* p; ~Guard() { delete p; } } guard = {jp};
	JFile& j = *jp;
	if (!j.decode()) { err = j.err; return false; }
	w = j.W; h = j.H;
	const int n = j.nc >= 3 ? 3 : 1;
	comp = n;
	const bool isRgb = j.nc == 3 && (j.rgbIds == 3 || (j.adobe == 0 && !j.jfif));
	const int decodeN = j.nc; // req_comp = 0: every stored component is resampled
	struct Up { int hs, vs, ystep, wlo, ypos; const u8 *line0, *line1; std::vector<u8> buf; } up[4];
	for (int k = 0; k < decodeN; ++k) {
		Up& u = up[k];
		u.hs = j.hmax / j.c[k].h; u.vs = j.vmax / j.c[k].v;
		u.ystep = u.vs >> 1;
		u.wlo = (j.W + u.hs - 1) / u.hs;
		u.ypos = 0;
		u.line0 = u.line1 = j.c[k].plane.data();
		u.buf.assign((size_t)j.W + 3 + 2 * u.hs, 0);
	}
	out.resize((size_t)w * h * n);
	for (int y = 0; y < h; ++y) {
		u8* o = &out[(size_t)n * w * y];
		const u8* row[4] = {NULL, NULL, NULL, NULL};
		for (int k = 0; k < decodeN; ++k) {
			Up& u = up[k];
			const bool bottom = u.ystep >= (u.vs >> 1);
			row[k] = upsampleRow(u.buf.data(), bottom ? u.line1 : u.line0, bottom ? u.line0 : u.line1, u.wlo, u.hs, u.vs);
			if (++u.ystep >= u.vs) {
				u.ystep = 0;
				u.line0 = u.line1;
				if (++u.ypos < j.c[k].ph) u.line1 += j.c[k].w2;
			}
		}
		if (n == 1) { memcpy(o, row[0], (size_t)w); continue; }
		if (j.nc == 3) {
			if (isRgb) for (int i = 0; i < w; ++i, o += 3) { o[0] = row[0][i]; o[1] = row[1][i]; o[2] = row[2][i]; }
			else yccToRgbRow(o, row[0], row[1], row[2], w, 3);
		} else { // four components: CMYK, YCCK, or YCbCr + an ignored fourth
			if (j.adobe == 0) for (int i = 0; i < w; ++i, o += 3) { const int m = row[3][i]; o[0] = mul255(row[0][i], m); o[1] = mul255(row[1][i], m); o[2] = mul255(row[2][i], m); }
			else {
				yccToRgbRow(o, row[0], row[1], row[2], w, 3);
				if (j.adobe == 2) for (int i = 0; i < w; ++i, o += 3) { const int m = row[3][i]; o[0] = mul255(255 - o[0], m); o[1] = mul255(255 - o[1], m); o[2] = mul255(255 - o[2], m); }
			}
		}
	}
	return true;
}

} // namespace

// Format detection in the reference decoder's order (reference: external/include/stb/stb_image.h:956-999): JPEG,
// PNG, then BMP / GIF / PSD / PIC / PNM / HDR (recognised here only to be refused by name), TGA last.
bool loadImage8(const std::string& path, int& w, int& h, int& comp, std::vector<unsigned char>& pixels, std::string& err) {
	std::vector<u8> f;
	if (!readFile(path, f)) { err = "cannot read " + path; return false; }
	static const u8 pngSig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
	if (f.size() >= 2 && f[0] == 0xFF && f[1] == 0xD8) return decodeJPEG(f, w, h, comp, pixels, err);
	if (f.size() >= 8 && !memcmp(f.data(), pngSig, 8)) return decodePNG(f, w, h, comp, pixels, err);
	const char* other = NULL;
	if (f.size() >= 2 && f[0] == 'B' && f[1] == 'M') other = "BMP";
	else if (f.size() >= 6 && (!memcmp(f.data(), "GIF87a", 6) || !memcmp(f.data(), "GIF89a", 6))) other = "GIF";
	else if (f.size() >= 4 && !memcmp(f.data(), "8BPS", 4)) other = "PSD";
	else if (f.size() >= 4 && f[0] == 0x53 && f[1] == 0x80 && f[2] == 0xF6 && f[3] == 0x34) other = "PIC";
	else if (f.size() >= 2 && f[0] == 'P' && (f[1] == '5' || f[1] == '6')) other = "PNM";
	else if (f.size() >= 6 && (!memcmp(f.data(), "#?RADI", 6) || !memcmp(f.data(), "#?RGBE", 6))) other = "Radiance HDR";
	if (other) { err = std::string(other) + " images are not supported: " + path; return false; }
	if (looksLikeTGA(f)) return decodeTGA(f, w, h, comp, pixels, err);
	err = "unknown image type: " + path;
	return false;
}

bool loadImageFloat(const std::string& path, int& w, int& h, int& comp, std::vector<float>& pixels, std::string& err) {
	std::vector<u8> px;
	if (!loadImage8(path, w, h, comp, px, err)) return false;
	// stbi__ldr_to_hdr in a C++ translation unit (as src/world_load.cpp:9-10 compiles it): pow(float, float) is the
	// float overload, gamma 2.2f, scale 1.0f; the alpha channel (last channel of an even count) is v / 255.0f.
	static float lut[256];
	static bool lutInit = false;
	if (!lutInit) {
		const float gamma = 2.2f, scale = 1.0f;
		for (int i = 0; i < 256; ++i) lut[i] = (float)(pow(i / 255.0f, gamma) * scale);
		lutInit = true;
	}
	const int n = (comp & 1) ? comp : comp - 1;
	pixels.resize(px.size());
	const size_t npix = (size_t)w * h;
	for (size_t i = 0; i < npix; ++i) {
		int k = 0;
		for (; k < n; ++k) pixels[i * comp + k] = lut[px[i * comp + k]];
		if (k < comp) pixels[i * comp + k] = px[i * comp + k] / 255.0f;
	}
	return true;
}

bool writePNG(const std::string& path, int w, int h, int comp, const unsigned char* pixels, std::string& err) {
	if (!(comp == 3 || comp == 4) || w <= 0 || h <= 0) { err = "writePNG: bad arguments"; return false; }
	const size_t stride = (size_t)w * comp;
	std::vector<u8> raw((stride + 1) * (size_t)h);
	for (int y = 0; y < h; ++y) {
		raw[(stride + 1) * (size_t)y] = 0;
		memcpy(&raw[(stride + 1) * (size_t)y + 1], pixels + stride * (size_t)y, stride);
	}
	uLongf clen = compressBound((uLong)raw.size());
	std::vector<u8> z(clen);
	if (compress2(z.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "writePNG: deflate failed"; return false; }
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) { err = "cannot write " + path; return false; }
	static const u8 sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
	fwrite(sig, 1, 8, f);
	auto chunk = [&](const char* type, const u8* data, unsigned len) {
		u8 hdr[8] = {(u8)(len >> 24), (u8)(len >> 16), (u8)(len >> 8), (u8)len, (u8)type[0], (u8)type[1], (u8)type[2], (u8)type[3]};
		fwrite(hdr, 1, 8, f);
		if (len) fwrite(data, 1, len, f);
		uLong crc = crc32(0L, Z_NULL, 0);
		crc = crc32(crc, hdr + 4, 4);
		if (len) crc = crc32(crc, data, len);
		u8 c[4] = {(u8)(crc >> 24), (u8)(crc >> 16), (u8)(crc >> 8), (u8)crc};
		fwrite(c, 1, 4, f);
	};
	u8 ihdr[13] = {(u8)(w >> 24), (u8)(w >> 16), (u8)(w >> 8), (u8)w, (u8)(h >> 24), (u8)(h >> 16), (u8)(h >> 8), (u8)h,
	               8, (u8)(comp == 3 ? 2 : 6), 0, 0, 0};
	chunk("IHDR", ihdr, 13);
	chunk("IDAT", z.data(), (unsigned)clen);
	chunk("IEND", NULL, 0);
	fclose(f);
	return true;
}

bool writePFM(const std::string& path, int w, int h, const float* rgb, std::string& err) {
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) { err = "cannot write " + path; return false; }
	fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
	for (int y = h - 1; y >= 0; --y) fwrite(rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f);
	fclose(f);
	return true;
}
