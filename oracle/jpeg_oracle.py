"""CPU restatement of tf.image.adjust_jpeg_quality / random_jpeg_quality (reference ModelComponents/pipeline.py:319-325) - TEST INFRASTRUCTURE,
never imported by the product path.

TensorFlow implements it as convert_image_dtype(uint8, saturate) -> encode_jpeg(quality, chroma down-sampling, libjpeg defaults) ->
decode_jpeg(defaults: slow integer IDCT, fancy up-sampling) -> convert_image_dtype(float).  The codec is a third-party dependency that is
absent from /root/reference: libjpeg-turbo (bundled with TensorFlow; version unpinned by the reference).  Entropy coding is lossless, so
the round trip is: colour conversion, 2x2 chroma down-sampling, 8x8 integer forward DCT, quantisation with the quality-scaled Annex K
tables, de-quantisation, integer inverse DCT, "fancy" (triangle) chroma up-sampling, colour conversion back - restated here from the
published algorithm (function names of libjpeg's sources in the docstrings).  PINNED: tests/test_jpeg_quality.py checks it bit for bit
against a real libjpeg-turbo (Pillow's encoder + decoder, the same library family TensorFlow bundles) on random and ragged images."""
import numpy as np

STD_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                     18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], np.int64).reshape(8, 8)
STD_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                       99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99], np.int64).reshape(8, 8)

CONST_BITS, PASS1_BITS = 13, 2
F_0_298, F_0_390, F_0_541, F_0_765, F_0_899, F_1_175 = 2446, 3196, 4433, 6270, 7373, 9633
F_1_501, F_1_847, F_1_961, F_2_053, F_2_562, F_3_072 = 12299, 15137, 16069, 16819, 20995, 25172


def quant_table(base: np.ndarray, quality: int) -> np.ndarray:
    """jpeg_quality_scaling + jpeg_add_quant_table(force_baseline=TRUE) (jcparam.c)."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    return np.clip((base * scale + 50) // 100, 1, 255)


def descale(x, n):
    return (x + (1 << (n - 1))) >> n


def rgb_to_ycc(rgb: np.ndarray):
    """rgb_ycc_convert (jccolor.c): 16-bit fixed point, Cb / Cr biased by ONE_HALF - 1."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16
    return y, cb, cr


def pad_edge(p: np.ndarray, h: int, w: int) -> np.ndarray:
    """expand_right_edge / expand_bottom_edge (jcsample.c, jcprepct.c): replicate the last column / row."""
    return np.pad(p, ((0, h - p.shape[0]), (0, w - p.shape[1])), mode="edge")


def downsample_h2v2(p: np.ndarray) -> np.ndarray:
    """h2v2_downsample (jcsample.c): 2x2 box filter, rounding bias alternating 1, 2 along a row."""
    s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)[None, :]
    return (s + bias) >> 2


def fdct_islow(block: np.ndarray) -> np.ndarray:
    """jpeg_fdct_islow (jfdctint.c) on level-shifted samples; output scaled by 8."""
    d = block.astype(np.int64)

    def one_pass(d, first):
        t0, t7 = d[..., 0] + d[..., 7], d[..., 0] - d[..., 7]
        t1, t6 = d[..., 1] + d[..., 6], d[..., 1] - d[..., 6]
        t2, t5 = d[..., 2] + d[..., 5], d[..., 2] - d[..., 5]
        t3, t4 = d[..., 3] + d[..., 4], d[..., 3] - d[..., 4]
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        o = [None] * 8
        if first:
            o[0], o[4] = (t10 + t11) << PASS1_BITS, (t10 - t11) << PASS1_BITS
            sh = CONST_BITS - PASS1_BITS
        else:
            o[0], o[4] = descale(t10 + t11, PASS1_BITS), descale(t10 - t11, PASS1_BITS)
            sh = CONST_BITS + PASS1_BITS
        z1 = (t12 + t13) * F_0_541
        o[2] = descale(z1 + t13 * F_0_765, sh)
        o[6] = descale(z1 - t12 * F_1_847, sh)
        z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
        z5 = (z3 + z4) * F_1_175
        t4, t5, t6, t7 = t4 * F_0_298, t5 * F_2_053, t6 * F_3_072, t7 * F_1_501
        z1, z2, z3, z4 = -z1 * F_0_899, -z2 * F_2_562, -z3 * F_1_961 + z5, -z4 * F_0_390 + z5
        o[7], o[5], o[3], o[1] = descale(t4 + z1 + z3, sh), descale(t5 + z2 + z4, sh), descale(t6 + z2 + z3, sh), descale(t7 + z1 + z4, sh)
        return np.stack(o, axis=-1)

    rows = one_pass(d, True)                                   # pass 1: rows
    cols = one_pass(np.swapaxes(rows, -1, -2), False)          # pass 2: columns
    return np.swapaxes(cols, -1, -2)


def quantize(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """quantize (jcdctmgr.c): divisor = 8 * table entry, round half away from zero."""
    div = qt * 8
    a = np.abs(coef)
    return np.sign(coef) * ((a + (div >> 1)) // div)


def idct_islow(coef: np.ndarray) -> np.ndarray:
    """jpeg_idct_islow (jidctint.c) on de-quantised coefficients; returns samples clamped to 0..255."""
    c = coef.astype(np.int64)

    def one_pass(c, first):
        z2, z3 = c[..., 2], c[..., 6]
        z1 = (z2 + z3) * F_0_541
        t2, t3 = z1 - z3 * F_1_847, z1 + z2 * F_0_765
        z2, z3 = c[..., 0], c[..., 4]
        t0, t1 = (z2 + z3) << CONST_BITS, (z2 - z3) << CONST_BITS
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        t0, t1, t2, t3 = c[..., 7], c[..., 5], c[..., 3], c[..., 1]
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F_1_175
        t0, t1, t2, t3 = t0 * F_0_298, t1 * F_2_053, t2 * F_3_072, t3 * F_1_501
        z1, z2, z3, z4 = -z1 * F_0_899, -z2 * F_2_562, -z3 * F_1_961 + z5, -z4 * F_0_390 + z5
        t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
        sh = CONST_BITS - PASS1_BITS if first else CONST_BITS + PASS1_BITS + 3
        o = [descale(t10 + t3, sh), descale(t11 + t2, sh), descale(t12 + t1, sh), descale(t13 + t0, sh),
             descale(t13 - t0, sh), descale(t12 - t1, sh), descale(t11 - t2, sh), descale(t10 - t3, sh)]
        return np.stack(o, axis=-1)

    cols = one_pass(np.swapaxes(c, -1, -2), True)              # pass 1: columns
    rows = one_pass(np.swapaxes(cols, -1, -2), False)          # pass 2: rows
    return np.clip(rows + 128, 0, 255)


def codec_plane(p: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """forward DCT + quantise + de-quantise + inverse DCT of a plane whose sides are multiples of 8."""
    h, w = p.shape
    b = (p - 128).reshape(h // 8, 8, w // 8, 8).swapaxes(1, 2)
    q = quantize(fdct_islow(b), qt)
    r = idct_islow(q * qt)
    return r.swapaxes(1, 2).reshape(h, w)


def upsample_h2v2_fancy(c: np.ndarray, H: int, W: int) -> np.ndarray:
    """h2v2_fancy_upsample (jdsample.c): triangle filter, 3/4 nearer + 1/4 further sample in each direction; the rows above the first and
    below the last REAL chroma row, and the columns beyond the real chroma width, do not exist (edges use the nearer sample alone)."""
    ch, cw = (H + 1) // 2, (W + 1) // 2
    c = c[:ch, :cw].astype(np.int64)
    if cw <= 2:                                                 # jinit_upsampler: the fancy filter needs more than two chroma columns, else plain replication
        return np.repeat(np.repeat(c, 2, axis=0), 2, axis=1)[:H, :W]
    up, dn = np.vstack([c[:1], c[:-1]]), np.vstack([c[1:], c[-1:]])
    out = np.zeros((2 * ch, 2 * cw), np.int64)
    for v, far in ((0, up), (1, dn)):
        this = 3 * c + far                                      # column sums of the output row pair
        last = np.hstack([this[:, :1], this[:, :-1]])
        nxt = np.hstack([this[:, 1:], this[:, -1:]])
        even = (3 * this + last + 8) >> 4
        odd = (3 * this + nxt + 7) >> 4
        even[:, 0] = (this[:, 0] * 4 + 8) >> 4                  # first column
        odd[:, -1] = (this[:, -1] * 4 + 7) >> 4                 # last column
        out[v::2, 0::2], out[v::2, 1::2] = even, odd
    return out[:H, :W]


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    """ycc_rgb_convert (jdcolor.c)."""
    cbx, crx = cb - 128, cr - 128
    r = y + ((91881 * crx + 32768) >> 16)
    b = y + ((116130 * cbx + 32768) >> 16)
    g = y + ((-22554 * cbx - 46802 * crx + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def jpeg_quality_roundtrip_u8(img: np.ndarray, quality: int) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [H, W, 3] through a quality-`quality` baseline JPEG with 4:2:0 chroma (libjpeg defaults)."""
    H, W, _ = img.shape
    y, cb, cr = rgb_to_ycc(img)
    H16, W16, H2 = -(-H // 16) * 16, -(-W // 16) * 16, -(-H // 2) * 2
    # Columns are replicated BEFORE the down-sampling (expand_right_edge on the input), rows only up to the next even row; the rest of the
    # bottom MCU row is filled by replicating the last DOWN-SAMPLED row (pre_process_data, jcprepct.c) - the two differ for even H.
    y = pad_edge(y, H16, W16)
    ql, qc = quant_table(STD_LUMA, quality), quant_table(STD_CHROMA, quality)
    y2 = codec_plane(y, ql)
    cb2, cr2 = (codec_plane(pad_edge(downsample_h2v2(pad_edge(c, H2, W16)), H16 // 2, W16 // 2), qc) for c in (cb, cr))
    return ycc_to_rgb(y2[:H, :W], upsample_h2v2_fancy(cb2, H, W), upsample_h2v2_fancy(cr2, H, W))


def adjust_jpeg_quality(image: np.ndarray, quality: int) -> np.ndarray:
    """tf.image.adjust_jpeg_quality for a float image in [0, 1]: saturating conversion to uint8 (x * 255.5 truncated), round trip, / 255."""
    u8 = np.clip(np.floor(image.astype(np.float64) * 255.5), 0, 255).astype(np.uint8)
    return (jpeg_quality_roundtrip_u8(u8, quality).astype(np.float32) / np.float32(255.0)).astype(np.float32)
