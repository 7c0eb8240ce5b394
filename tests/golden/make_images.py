#!/usr/bin/env python3
"""Writes tests/golden/img/*: small synthetic JPEG / PNG / TGA files that walk the decoder's branches
(chroma subsampling 4:4:4 / 4:2:2 / 4:2:0, progressive and restart-interval JPEGs, grey and CMYK JPEGs; PNG
colour types x bit depths x Adam7, tRNS, palettes; TGA raw / RLE / colour-mapped / 15-16 bit / either origin).
They are DATA made here with Pillow and by hand; nothing comes from the reference. make_golden.py then
freezes what the reference's own stb_image.h decodes them to (ref_pins.json, key "images").

    python tests/golden/make_images.py        (needs Pillow; the committed files are what the tests read)
"""
import os
import struct
import zlib

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "img")
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(7)


def P(name):
    return os.path.join(OUT, name)


def pic(w, h, c):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 7) % 256, 255 - (x * 3) % 256], -1)[..., :c].astype(np.int32)
    return np.clip(base + rng.integers(-40, 40, size=(h, w, c)), 0, 255).astype(np.uint8)


for (w, h) in [(1, 1), (7, 5), (33, 17)]:
    rgb = Image.fromarray(pic(w, h, 3), "RGB")
    g = Image.fromarray(pic(w, h, 1)[..., 0], "L")
    for ss, name in [(0, "444"), (1, "422"), (2, "420")]:
        rgb.save(P(f"j_{w}x{h}_{name}.jpg"), quality=85, subsampling=ss)
        rgb.save(P(f"jp_{w}x{h}_{name}.jpg"), quality=70, subsampling=ss, progressive=True)
        rgb.save(P(f"jr_{w}x{h}_{name}.jpg"), quality=95, subsampling=ss, restart_marker_blocks=3)
    g.save(P(f"jg_{w}x{h}.jpg"), quality=80)
    g.save(P(f"jgp_{w}x{h}.jpg"), quality=80, progressive=True)
    Image.fromarray(pic(w, h, 4), "CMYK").save(P(f"jc_{w}x{h}.jpg"), quality=80)
    rgb.save(P(f"jo_{w}x{h}.jpg"), quality=60, optimize=True)
    rgb.save(P(f"p_rgb_{w}x{h}.png"))
    Image.fromarray(pic(w, h, 4), "RGBA").save(P(f"p_rgba_{w}x{h}.png"))
    Image.fromarray(pic(w, h, 2), "LA").save(P(f"p_la_{w}x{h}.png"))
    rgb.quantize(17).save(P(f"p_pal_{w}x{h}.png"))
    rgb.quantize(4).save(P(f"p_pal4_{w}x{h}.png"), bits=2)
    rgb.quantize(9).save(P(f"p_palt_{w}x{h}.png"), transparency=bytes([0, 128, 255]))
    g.save(P(f"p_gt_{w}x{h}.png"), transparency=int(np.asarray(g)[0, 0]))
    rgb.save(P(f"p_rgbt_{w}x{h}.png"), transparency=tuple(int(v) for v in np.asarray(rgb)[0, 0]))
    g.point(lambda v: 255 if v > 127 else 0).convert("1").save(P(f"p_1_{w}x{h}.png"))
    rgb.save(P(f"t_rgb_{w}x{h}.tga"))
    rgb.save(P(f"t_rgbr_{w}x{h}.tga"), compression="tga_rle")
    Image.fromarray(pic(w, h, 4), "RGBA").save(P(f"t_rgba_{w}x{h}.tga"), compression="tga_rle")
    g.save(P(f"t_gr_{w}x{h}.tga"), compression="tga_rle")
    rgb.quantize(33).save(P(f"t_pal_{w}x{h}.tga"))
    rgb.quantize(33).save(P(f"t_palr_{w}x{h}.tga"), compression="tga_rle")
    rgb.save(P(f"t_top_{w}x{h}.tga"), orientation=1)


def chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))


def png(name, w, h, depth, ctype, px, interlace=0, extra=b""):
    """Hand-made PNG: every scanline filter type in turn, optional Adam7."""
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]

    def enc(sub):
        out, prev = b"", None
        for y in range(sub.shape[0]):
            vals = [int(v) for v in sub[y].reshape(-1)]
            if depth == 16:
                line = b"".join(struct.pack(">H", v) for v in vals)
            elif depth == 8:
                line = bytes(vals)
            else:
                bits = "".join(format(v, "0%db" % depth) for v in vals)
                bits += "0" * ((-len(bits)) % 8)
                line = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
            ft, bpp = y % 5, max(1, nch * depth // 8)
            if prev is None:
                prev = bytes(len(line))
            f = bytearray(len(line))
            for i in range(len(line)):
                a = line[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 0:
                    p = 0
                elif ft == 1:
                    p = a
                elif ft == 2:
                    p = b
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pp = a + b - c
                    pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                f[i] = (line[i] - p) & 255
            prev = line
            out += bytes([ft]) + bytes(f)
        return out

    if not interlace:
        raw = enc(px)
    else:
        raw = b""
        for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += enc(sub)
    with open(P(name), "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra
                 + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


for (w, h) in [(5, 3), (9, 9)]:
    for depth, ctype in [(1, 0), (2, 0), (4, 0), (8, 0), (16, 0), (8, 2), (16, 2), (8, 4), (16, 4), (8, 6), (16, 6)]:
        nch = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
        for il in (0, 1):
            png(f"h_{depth}_{ctype}_{il}_{w}x{h}.png", w, h, depth, ctype, rng.integers(0, 1 << depth, size=(h, w, nch)), il)
    png(f"h_t2_{w}x{h}.png", w, h, 2, 0, rng.integers(0, 4, size=(h, w, 1)), 0, chunk(b"tRNS", struct.pack(">H", 2)))
    png(f"h_t16_{w}x{h}.png", w, h, 16, 2, rng.integers(0, 2, size=(h, w, 3)) * 65535, 1, chunk(b"tRNS", struct.pack(">HHH", 65535, 0, 65535)))


def tga(name, w, h, type_, bits, data, desc=0, cmap=b"", cmaplen=0, cmapbits=0, idl=b""):
    with open(P(name), "wb") as fh:
        fh.write(struct.pack("<BBBHHBHHHHBB", len(idl), 1 if cmap else 0, type_, 0, cmaplen, cmapbits, 0, 0, w, h, bits, desc) + idl + cmap + data)


w, h = 13, 7
tga("m_16.tga", w, h, 2, 16, rng.integers(0, 65536, size=w * h).astype("<u2").tobytes())
tga("m_15.tga", w, h, 2, 15, rng.integers(0, 32768, size=w * h).astype("<u2").tobytes(), 0x20)
tga("m_g16.tga", w, h, 3, 16, rng.integers(0, 256, size=w * h * 2).astype("u1").tobytes(), 0, b"", 0, 0, b"hello")
tga("m_p15.tga", w, h, 1, 8, rng.integers(0, 20, size=w * h).astype("u1").tobytes(), 0, rng.integers(0, 65536, size=16).astype("<u2").tobytes(), 16, 15)
rle, n = b"", w * h
while n > 0:
    k = min(n, int(rng.integers(1, 20)))
    rep = int(rng.integers(0, 2))
    rle += bytes([(rep << 7) | (k - 1)]) + (bytes(rng.integers(0, 256, size=3).astype("u1")) if rep else bytes(rng.integers(0, 256, size=3 * k).astype("u1")))
    n -= k
tga("m_rle24.tga", w, h, 10, 24, rle, 0x20)
# damaged files: both decoders must agree on "does not load" / on what a partial stream decodes to
for f in ["j_33x17_420.jpg", "jp_33x17_422.jpg", "p_rgb_33x17.png", "t_rgbr_33x17.tga"]:
    b = open(P(f), "rb").read()
    open(P("cut_" + f), "wb").write(b[:len(b) * 2 // 3])
    bb = bytearray(b)
    bb[len(b) * 3 // 4] ^= 0x10
    open(P("flip_" + f), "wb").write(bytes(bb))
print(len(os.listdir(OUT)), "files,", sum(os.path.getsize(P(f)) for f in os.listdir(OUT)), "bytes")
