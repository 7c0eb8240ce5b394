// PNG / TGA / baseline-JPEG decoders and PNG / PFM writers. See image_io.h.
#include "image_io.h"

#include <zlib.h>

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

// ---------------------------------------------------------------- PNG
int paeth(int a, int b, int c) {
	int p = a + b - c;
	int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
	if (pa <= pb && pa <= pc) return a;
	if (pb <= pc) return b;
	return c;
}

bool decodePNG(const std::vector<u8>& f, int& w, int& h, int& comp, std::vector<u8>& out, std::string& err) {
	static const u8 sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
	if (f.size() < 8 || memcmp(f.data(), sig, 8)) { err = "not a PNG"; return false; }
	size_t p = 8;
	int depth = 0, ctype = 0, interlace = 0;
	std::vector<u8> idat, plte, trns;
	bool gotHdr = false;
	while (p + 8 <= f.size()) {
		unsigned len = be32(&f[p]);
		const u8* type = &f[p + 4];
		const u8* data = &f[p + 8];
		if (p + 12 + (size_t)len > f.size()) { err = "PNG: truncated chunk"; return false; }
		if (!memcmp(type, "IHDR", 4)) {
			if (len < 13) { err = "PNG: bad IHDR"; return false; }
			w = (int)be32(data); h = (int)be32(data + 4);
			depth = data[8]; ctype = data[9]; interlace = data[12];
			gotHdr = true;
		} else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
		else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
		else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
		else if (!memcmp(type, "IEND", 4)) break;
		p += 12 + (size_t)len;
	}
	if (!gotHdr || w <= 0 || h <= 0) { err = "PNG: missing IHDR"; return false; }
	if (interlace) { err = "PNG: interlaced files are not supported"; return false; }
	int nch;
	switch (ctype) {
		case 0: nch = 1; break;
		case 2: nch = 3; break;
		case 3: nch = 1; break;
		case 4: nch = 2; break;
		case 6: nch = 4; break;
		default: err = "PNG: bad colour type"; return false;
	}
	if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
		err = "PNG: unsupported bit depth"; return false;
	}
	const size_t bpp_bits = (size_t)nch * depth;
	const size_t stride = ((size_t)w * bpp_bits + 7) / 8;
	const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
	std::vector<u8> raw((stride + 1) * (size_t)h);
	uLongf rawLen = (uLongf)raw.size();
	int zr = uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size());
	if (zr != Z_OK || rawLen != raw.size()) { err = "PNG: inflate failed"; return false; }
	// unfilter in place
	std::vector<u8> img(stride * (size_t)h);
	for (int y = 0; y < h; ++y) {
		const u8* src = &raw[(stride + 1) * (size_t)y];
		u8* cur = &img[stride * (size_t)y];
		const u8* up = y ? &img[stride * (size_t)(y - 1)] : NULL;
		int ft = src[0];
		++src;
		for (size_t x = 0; x < stride; ++x) {
			int a = x >= bpp ? cur[x - bpp] : 0;
			int b = up ? up[x] : 0;
			int c = (up && x >= bpp) ? up[x - bpp] : 0;
			int v = src[x];
			switch (ft) {
				case 0: break;
				case 1: v += a; break;
				case 2: v += b; break;
				case 3: v += (a + b) >> 1; break;
				case 4: v += paeth(a, b, c); break;
				default: err = "PNG: bad filter"; return false;
			}
			cur[x] = (u8)v;
		}
	}
	// expand to 8-bit channels
	bool pal = ctype == 3;
	bool palAlpha = pal && !trns.empty();
	comp = pal ? (palAlpha ? 4 : 3) : nch;
	out.resize((size_t)w * h * comp);
	for (int y = 0; y < h; ++y) {
		const u8* row = &img[stride * (size_t)y];
		for (int x = 0; x < w; ++x) {
			u8 v[4] = {0, 0, 0, 255};
			for (int c = 0; c < nch; ++c) {
				unsigned s;
				if (depth == 8) s = row[(size_t)x * nch + c];
				else if (depth == 16) s = row[((size_t)x * nch + c) * 2]; // high byte
				else {
					size_t bit = (size_t)x * depth;
					s = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
					if (!pal) s = s * 255 / ((1u << depth) - 1);
				}
				v[c] = (u8)s;
			}
			u8* o = &out[((size_t)y * w + x) * comp];
			if (pal) {
				unsigned idx = v[0];
				o[0] = idx * 3 + 2 < plte.size() ? plte[idx * 3] : 0;
				o[1] = idx * 3 + 2 < plte.size() ? plte[idx * 3 + 1] : 0;
				o[2] = idx * 3 + 2 < plte.size() ? plte[idx * 3 + 2] : 0;
				if (palAlpha) o[3] = idx < trns.size() ? trns[idx] : 255;
			} else {
				for (int c = 0; c < nch; ++c) o[c] = v[c];
			}
		}
	}
	return true;
}

// ---------------------------------------------------------------- TGA
bool decodeTGA(const std::vector<u8>& f, int& w, int& h, int& comp, std::vector<u8>& out, std::string& err) {
	if (f.size() < 18) { err = "TGA: too short"; return false; }
	int idLen = f[0], cmapType = f[1], type = f[2];
	int cmapLen = f[5] | (f[6] << 8), cmapBits = f[7];
	w = f[12] | (f[13] << 8);
	h = f[14] | (f[15] << 8);
	int bpp = f[16], desc = f[17];
	bool rle = type >= 8;
	int base = type & 7;
	if (cmapType != 0 || !(base == 2 || base == 3)) { err = "TGA: only true-colour / grayscale images are supported"; return false; }
	if (!(bpp == 8 || bpp == 24 || bpp == 32) || w <= 0 || h <= 0) { err = "TGA: unsupported pixel depth"; return false; }
	comp = bpp / 8;
	size_t p = 18 + (size_t)idLen + (size_t)cmapLen * ((cmapBits + 7) / 8);
	out.resize((size_t)w * h * comp);
	size_t npix = (size_t)w * h, i = 0;
	u8 px[4] = {0, 0, 0, 0};
	int rleCount = 0;
	bool rleRepeat = false;
	while (i < npix) {
		bool readPx = true;
		if (rle) {
			if (rleCount == 0) {
				if (p >= f.size()) { err = "TGA: truncated"; return false; }
				int c = f[p++];
				rleCount = 1 + (c & 127);
				rleRepeat = (c & 128) != 0;
			} else if (rleRepeat) readPx = false;
		}
		if (readPx) {
			if (p + comp > f.size()) { err = "TGA: truncated"; return false; }
			for (int c = 0; c < comp; ++c) px[c] = f[p++];
		}
		u8* o = &out[i * comp];
		if (comp >= 3) { o[0] = px[2]; o[1] = px[1]; o[2] = px[0]; if (comp == 4) o[3] = px[3]; } // BGR(A) -> RGB(A)
		else o[0] = px[0];
		++i;
		if (rle) --rleCount;
	}
	if (!(desc & 0x20)) { // bottom-left origin -> flip to top-left
		size_t rowb = (size_t)w * comp;
		std::vector<u8> tmp(rowb);
		for (int y = 0; y < h / 2; ++y) {
			u8* a = &out[rowb * (size_t)y];
			u8* b = &out[rowb * (size_t)(h - 1 - y)];
			memcpy(tmp.data(), a, rowb); memcpy(a, b, rowb); memcpy(b, tmp.data(), rowb);
		}
	}
	return true;
}

// ---------------------------------------------------------------- JPEG (baseline / sequential Huffman)
struct Huff {
	u8 bits[17];
	u8 vals[256];
	int mincode[17], maxcode[18], valptr[17];
	bool ok;
	Huff() : ok(false) {}
	void build() {
		int code = 0, k = 0;
		for (int l = 1; l <= 16; ++l) {
			valptr[l] = k;
			mincode[l] = code;
			code += bits[l];
			k += bits[l];
			maxcode[l] = bits[l] ? code - 1 : -1;
			code <<= 1;
		}
		maxcode[17] = 0x7fffffff;
		ok = true;
	}
};

struct JComp { int id, h, v, tq, td, ta; int pred; int bw, bh; std::vector<u8> plane; };

struct JDec {
	const u8* d; size_t n, p;
	unsigned bitbuf; int bitcnt;
	bool eof;
	JDec() : d(0), n(0), p(0), bitbuf(0), bitcnt(0), eof(false) {}
	int getbit() {
		if (bitcnt == 0) {
			if (p >= n) { eof = true; return 0; }
			unsigned b = d[p++];
			if (b == 0xFF) {
				if (p < n && d[p] == 0) ++p;
				else { --p; eof = true; return 0; } // marker
			}
			bitbuf = b; bitcnt = 8;
		}
		--bitcnt;
		return (bitbuf >> bitcnt) & 1;
	}
	int getbits(int c) { int v = 0; while (c--) v = (v << 1) | getbit(); return v; }
	int decode(const Huff& h) {
		int code = 0;
		for (int l = 1; l <= 16; ++l) {
			code = (code << 1) | getbit();
			if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
		}
		return 0;
	}
	static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
};

const int kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// separable float IDCT from the definition (texture decode is load-time only)
void idct8x8(const int* coef, u8* out, int stride) {
	static float C[8][8];
	static bool init = false;
	if (!init) {
		for (int x = 0; x < 8; ++x)
			for (int u = 0; u < 8; ++u)
				C[x][u] = (u == 0 ? (float)(1.0 / sqrt(2.0)) : 1.0f) * (float)cos((2 * x + 1) * u * M_PI / 16.0) * 0.5f;
		init = true;
	}
	float tmp[64];
	for (int v = 0; v < 8; ++v)
		for (int x = 0; x < 8; ++x) {
			float s = 0;
			for (int u = 0; u < 8; ++u) s += C[x][u] * (float)coef[v * 8 + u];
			tmp[v * 8 + x] = s;
		}
	for (int y = 0; y < 8; ++y)
		for (int x = 0; x < 8; ++x) {
			float s = 0;
			for (int v = 0; v < 8; ++v) s += C[y][v] * tmp[v * 8 + x];
			int iv = (int)floorf(s + 128.5f);
			out[y * stride + x] = (u8)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
		}
}

bool decodeJPEG(const std::vector<u8>& f, int& w, int& h, int& comp, std::vector<u8>& out, std::string& err) {
	if (f.size() < 4 || f[0] != 0xFF || f[1] != 0xD8) { err = "not a JPEG"; return false; }
	size_t p = 2;
	int qt[4][64];
	memset(qt, 0, sizeof(qt));
	Huff hdc[4], hac[4];
	std::vector<JComp> comps;
	int restart = 0, hmax = 1, vmax = 1;
	w = h = 0;
	while (p + 4 <= f.size()) {
		if (f[p] != 0xFF) { ++p; continue; }
		int m = f[p + 1];
		p += 2;
		if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01 || m == 0xFF) { if (m == 0xFF) --p; continue; }
		if (m == 0xD9) break;
		unsigned len = be16(&f[p]);
		if (p + len > f.size()) { err = "JPEG: truncated segment"; return false; }
		const u8* s = &f[p + 2];
		unsigned sl = len - 2;
		if (m == 0xDB) {
			unsigned q = 0;
			while (q < sl) {
				int pq = s[q] >> 4, tq = s[q] & 15;
				++q;
				if (tq > 3) { err = "JPEG: bad DQT"; return false; }
				for (int i = 0; i < 64; ++i) {
					qt[tq][kZigzag[i]] = pq ? (int)be16(&s[q]) : s[q];
					q += pq ? 2 : 1;
				}
			}
		} else if (m == 0xC4) {
			unsigned q = 0;
			while (q + 17 <= sl) {
				int tc = s[q] >> 4, th = s[q] & 15;
				++q;
				if (th > 3) { err = "JPEG: bad DHT"; return false; }
				Huff& hf = tc ? hac[th] : hdc[th];
				int total = 0;
				hf.bits[0] = 0;
				for (int i = 1; i <= 16; ++i) { hf.bits[i] = s[q++]; total += hf.bits[i]; }
				if (total > 256 || q + total > sl) { err = "JPEG: bad DHT"; return false; }
				memcpy(hf.vals, &s[q], (size_t)total);
				q += total;
				hf.build();
			}
		} else if (m == 0xC0 || m == 0xC1) {
			if (s[0] != 8) { err = "JPEG: only 8-bit precision is supported"; return false; }
			h = (int)be16(&s[1]); w = (int)be16(&s[3]);
			int nc = s[5];
			if (!(nc == 1 || nc == 3)) { err = "JPEG: only 1 or 3 components are supported"; return false; }
			comps.resize((size_t)nc);
			for (int i = 0; i < nc; ++i) {
				comps[i].id = s[6 + 3 * i];
				comps[i].h = s[7 + 3 * i] >> 4;
				comps[i].v = s[7 + 3 * i] & 15;
				comps[i].tq = s[8 + 3 * i];
				if (comps[i].h < 1 || comps[i].h > 2 || comps[i].v < 1 || comps[i].v > 2 || comps[i].tq > 3) { err = "JPEG: unsupported sampling"; return false; }
				if (comps[i].h > hmax) hmax = comps[i].h;
				if (comps[i].v > vmax) vmax = comps[i].v;
			}
		} else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
			err = "JPEG: progressive / lossless / arithmetic files are not supported";
			return false;
		} else if (m == 0xDD) {
			restart = (int)be16(s);
		} else if (m == 0xDA) {
			if (comps.empty() || w <= 0 || h <= 0) { err = "JPEG: SOS before SOF"; return false; }
			int ns = s[0];
			if (ns != (int)comps.size()) { err = "JPEG: non-interleaved scans are not supported"; return false; }
			for (int i = 0; i < ns; ++i) {
				int cid = s[1 + 2 * i];
				for (size_t c = 0; c < comps.size(); ++c)
					if (comps[c].id == cid) { comps[c].td = s[2 + 2 * i] >> 4; comps[c].ta = s[2 + 2 * i] & 15; }
			}
			p += len;
			const int mcuw = 8 * hmax, mcuh = 8 * vmax;
			const int mx = (w + mcuw - 1) / mcuw, my = (h + mcuh - 1) / mcuh;
			for (size_t c = 0; c < comps.size(); ++c) {
				comps[c].bw = mx * comps[c].h * 8;
				comps[c].bh = my * comps[c].v * 8;
				comps[c].plane.assign((size_t)comps[c].bw * comps[c].bh, 0);
				comps[c].pred = 0;
				if (!hdc[comps[c].td & 3].ok || !hac[comps[c].ta & 3].ok) { err = "JPEG: missing Huffman table"; return false; }
			}
			JDec jd;
			jd.d = f.data(); jd.n = f.size(); jd.p = p;
			int mcuCount = 0;
			for (int yy = 0; yy < my; ++yy)
				for (int xx = 0; xx < mx; ++xx) {
					if (restart && mcuCount && (mcuCount % restart) == 0) {
						// skip to after the RSTn marker
						jd.bitcnt = 0; jd.eof = false;
						while (jd.p + 1 < jd.n && !(jd.d[jd.p] == 0xFF && jd.d[jd.p + 1] >= 0xD0 && jd.d[jd.p + 1] <= 0xD7)) ++jd.p;
						jd.p += 2;
						for (size_t c = 0; c < comps.size(); ++c) comps[c].pred = 0;
					}
					for (size_t c = 0; c < comps.size(); ++c) {
						JComp& jc = comps[c];
						for (int by = 0; by < jc.v; ++by)
							for (int bx = 0; bx < jc.h; ++bx) {
								int coef[64];
								memset(coef, 0, sizeof(coef));
								int t = jd.decode(hdc[jc.td & 3]);
								int diff = t ? JDec::extend(jd.getbits(t), t) : 0;
								jc.pred += diff;
								coef[0] = jc.pred * qt[jc.tq][0];
								for (int k = 1; k < 64;) {
									int rs = jd.decode(hac[jc.ta & 3]);
									int r = rs >> 4, sbits = rs & 15;
									if (sbits == 0) {
										if (r == 15) { k += 16; continue; }
										break;
									}
									k += r;
									if (k > 63) break;
									int zz = kZigzag[k];
									coef[zz] = JDec::extend(jd.getbits(sbits), sbits) * qt[jc.tq][zz];
									++k;
								}
								int ox = (xx * jc.h + bx) * 8, oy = (yy * jc.v + by) * 8;
								idct8x8(coef, &jc.plane[(size_t)oy * jc.bw + ox], jc.bw);
							}
					}
					++mcuCount;
				}
			// colour conversion with replicated chroma
			comp = (int)comps.size();
			out.resize((size_t)w * h * comp);
			for (int y = 0; y < h; ++y)
				for (int x = 0; x < w; ++x) {
					u8* o = &out[((size_t)y * w + x) * comp];
					if (comp == 1) { o[0] = comps[0].plane[(size_t)y * comps[0].bw + x]; continue; }
					float Y = 0, Cb = 0, Cr = 0;
					for (int c = 0; c < 3; ++c) {
						const JComp& jc = comps[c];
						int sx = x * jc.h / hmax, sy = y * jc.v / vmax;
						float val = jc.plane[(size_t)sy * jc.bw + sx];
						if (c == 0) Y = val; else if (c == 1) Cb = val - 128.0f; else Cr = val - 128.0f;
					}
					float r = Y + 1.402f * Cr, g = Y - 0.344136f * Cb - 0.714136f * Cr, b = Y + 1.772f * Cb;
					int ir = (int)floorf(r + 0.5f), ig = (int)floorf(g + 0.5f), ib = (int)floorf(b + 0.5f);
					o[0] = (u8)(ir < 0 ? 0 : (ir > 255 ? 255 : ir));
					o[1] = (u8)(ig < 0 ? 0 : (ig > 255 ? 255 : ig));
					o[2] = (u8)(ib < 0 ? 0 : (ib > 255 ? 255 : ib));
				}
			return true;
		}
		p += len;
	}
	err = "JPEG: no scan found";
	return false;
}

bool endsWith(const std::string& s, const char* suf) {
	size_t n = strlen(suf);
	if (s.size() < n) return false;
	for (size_t i = 0; i < n; ++i) {
		char a = s[s.size() - n + i], b = suf[i];
		if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
		if (a != b) return false;
	}
	return true;
}

} // namespace

bool loadImage8(const std::string& path, int& w, int& h, int& comp, std::vector<unsigned char>& pixels, std::string& err) {
	std::vector<u8> f;
	if (!readFile(path, f)) { err = "cannot read " + path; return false; }
	if (f.size() >= 8 && f[0] == 137 && f[1] == 'P') return decodePNG(f, w, h, comp, pixels, err);
	if (f.size() >= 2 && f[0] == 0xFF && f[1] == 0xD8) return decodeJPEG(f, w, h, comp, pixels, err);
	if (endsWith(path, ".tga")) return decodeTGA(f, w, h, comp, pixels, err);
	err = "unsupported image format: " + path;
	return false;
}

bool loadImageFloat(const std::string& path, int& w, int& h, int& comp, std::vector<float>& pixels, std::string& err) {
	std::vector<u8> px;
	if (!loadImage8(path, w, h, comp, px, err)) return false;
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
