#!/usr/bin/env python3
"""tools/jxl_write.py — a minimal JPEG XL codestream WRITER (test infrastructure, not part of the decode path).

libjxl's encoder never selects DCT128x128 / 256x256 / 64x128 / 128x64 / 128x256 / 256x128 varblocks and its public API cannot place splines, yet
every decoder has to take such files (the reference's libjxl does).  This writes small VarDCT frames — one 256x256 group, everything in one
section — with a varblock layout, quantised coefficients, LF samples, chroma-from-luma factors, a quant field, EPF sharpness and (optionally)
splines chosen by the caller, so that the reference decoder (oracle/_ref) can turn them into golden pixels (tests/golden/make_golden.py: w_*).
It is the product's host parser run backwards (jxl_coder_amd/csrc/host_format.inc, host_parse.cpp, host_bits.cpp; ISO/IEC 18181-1):
  signature, SizeHeader, ImageMetadata (all_default: 8-bit sRGB, XYB), FrameHeader, TOC (one entry),
  LfGlobal (splines, default LF dequantisation, quantiser, default block-context map, default colour correlation, no global MA tree),
  LfGroup (LF coefficients and HF metadata as Modular streams with their own MA tree), HfGlobal (default dequant matrices, natural orders),
  PassGroup (nonzero counts + coefficients).  Entropy coding: prefix codes only (one cluster per stream, HybridUintConfig(0, 0, 0)).
"""
import heapq
import numpy as np

COVERED_X = [1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32]
COVERED_Y = [1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16]


class BW:
    """bit writer, least significant bit first"""
    def __init__(self):
        self.acc, self.n = 0, 0

    def w(self, nbits, v):
        assert 0 <= v < (1 << nbits) or nbits == 0, (nbits, v)
        self.acc |= v << self.n
        self.n += nbits

    def bool(self, b):
        self.w(1, 1 if b else 0)

    def u32(self, v, spec):
        """spec: four (bits, offset); bits < 0 = the constant `offset`"""
        for sel, (b, o) in enumerate(spec):
            if (b < 0 and v == o) or (b >= 0 and o <= v < o + (1 << b)):
                self.w(2, sel)
                if b >= 0:
                    self.w(b, v - o)
                return
        raise ValueError((v, spec))

    def u64(self, v):
        if v == 0:
            self.w(2, 0)
        elif v <= 16:
            self.w(2, 1); self.w(4, v - 1)
        elif v <= 272:
            self.w(2, 2); self.w(8, v - 17)
        else:
            raise ValueError('u64 beyond 272 is not needed here')

    def f16(self, x):
        self.w(16, int(np.float16(x).view(np.uint16)))

    def align(self):
        self.n = (self.n + 7) & ~7

    def bytes(self):
        self.align()
        return self.acc.to_bytes(self.n // 8, "little")


def pack_signed(v):
    return 2 * v if v >= 0 else -2 * v - 1


def huffman_lengths(freq):
    """code lengths (<= 15) of a complete prefix code over the symbols with freq > 0 (at least two of them)"""
    syms = [s for s, f in enumerate(freq) if f > 0]
    assert len(syms) >= 2
    for _ in range(8):
        heap = [(freq[s], i, (s,)) for i, s in enumerate(syms)]
        heapq.heapify(heap)
        lens = {s: 0 for s in syms}
        k = len(heap)
        while len(heap) > 1:
            a = heapq.heappop(heap); b = heapq.heappop(heap)
            for s in a[2] + b[2]:
                lens[s] += 1
            heapq.heappush(heap, (a[0] + b[0], k, a[2] + b[2])); k += 1
        if max(lens.values()) <= 15:
            break
        freq = [max(1, f // 2 + 1) if f else 0 for f in freq]       # flatten and retry
    out = [0] * len(freq)
    for s, l in lens.items():
        out[s] = l
    return out


def canonical_codes(lens):
    """symbol -> (code, length); codes are read most significant bit first"""
    code, out = 0, {}
    for l in range(1, 16):
        for s, ls in enumerate(lens):
            if ls == l:
                out[s] = (code, l); code += 1
        code <<= 1
    return out


def put_code(bw, code, length):
    for i in range(length - 1, -1, -1):
        bw.w(1, (code >> i) & 1)


def write_prefix_code(bw, freq):
    """the prefix code of one cluster (host_bits.cpp: read_prefix_code) for token frequencies `freq`; returns symbol -> (code, length)"""
    alphabet = len(freq)
    if alphabet == 1:
        return {0: (0, 0)}
    used = [s for s, f in enumerate(freq) if f > 0]
    max_bits = (alphabet - 1).bit_length()
    if len(used) <= 1:                                   # simple code, one symbol: no bits per token
        bw.w(2, 1); bw.w(2, 0); bw.w(max_bits, used[0] if used else 0)
        return {(used[0] if used else 0): (0, 0)}
    if len(used) == 2:
        bw.w(2, 1); bw.w(2, 1); bw.w(max_bits, used[0]); bw.w(max_bits, used[1])
        return {used[0]: (0, 1), used[1]: (1, 1)}
    lens = huffman_lengths(freq)
    if len(set(l for l in lens[:used[-1] + 1])) == 1:      # one code-length value only: skew the code so that the code-length code has two symbols
        f2 = list(freq); f2[used[0]] = sum(freq) * 4
        lens = huffman_lengths(f2)
    seq = lens[:used[-1] + 1]
    clfreq = [0] * 18
    for l in seq:
        clfreq[l] += 1
    cll = huffman_lengths(clfreq)
    if max(cll) > 5:
        raise ValueError("code-length code too deep")
    bw.w(2, 0)                                             # HSKIP = 0
    order = [1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15]
    fixed = {0: (0, 2), 4: (1, 2), 3: (2, 2), 2: (3, 3), 1: (7, 4), 5: (15, 4)}       # value -> (bits as an LSB-first integer, length)
    space = 32
    for sym in order:
        if space <= 0:
            break
        v = cll[sym]
        bits, n = fixed[v]
        bw.w(n, bits)
        if v:
            space -= 32 >> v
    assert space == 0, space
    clc = canonical_codes(cll)
    sp = 32768
    for l in seq:
        if sp <= 0:
            break
        put_code(bw, *clc[l])
        if l:
            sp -= 32768 >> l
    assert sp == 0, sp
    return canonical_codes(lens)


class EC:
    """an entropy-coded stream with prefix codes: every context in ONE cluster, HybridUintConfig(0, 0, 0) (token = 1 + floor(log2 v))"""
    def __init__(self, num_ctx):
        self.num_ctx, self.syms = num_ctx, []

    def add(self, ctx, value):
        assert 0 <= ctx < self.num_ctx and value >= 0
        self.syms.append(value)

    @staticmethod
    def token(v):
        if v == 0:
            return 0, 0, 0
        n = v.bit_length() - 1
        return 1 + n, n, v - (1 << n)

    def write_header(self, bw):
        bw.bool(0)                                         # no LZ77
        if self.num_ctx > 1:
            bw.bool(1); bw.w(2, 0)                         # simple context map, 0 bits per entry: one cluster
        bw.bool(1)                                         # prefix codes (log_alpha_size 15)
        bw.w(4, 0)                                         # split_exponent 0 (msb_in_token / lsb_in_token: 0 bits each)
        toks = [self.token(v)[0] for v in self.syms]
        count = max(toks) + 1 if toks else 1
        if count == 1:
            bw.bool(0)
        else:
            bw.bool(1); nb = (count - 1).bit_length() - 1; bw.w(4, nb); bw.w(nb, count - 1 - (1 << nb))
        freq = [0] * count
        for t in toks:
            freq[t] += 1
        self.codes = write_prefix_code(bw, freq)

    def write_symbols(self, bw, first=0, last=None):
        for v in self.syms[first:last]:
            t, n, bits = self.token(v)
            put_code(bw, *self.codes[t])
            bw.w(n, bits)


# ---- MA trees: ("split", property, value, subtree if property > value, subtree otherwise) | ("leaf", predictor, offset)
def tree_bfs(tree):
    nodes, queue = [], [tree]
    while queue:
        nd = queue.pop(0)
        nodes.append(nd)
        if nd[0] == "split":
            queue.append(nd[3]); queue.append(nd[4])
    return nodes


def write_tree(bw, tree):
    """H.4.2: the tree in breadth-first order through a 6-context code, then the header of the leaves' symbol code is the caller's"""
    ec = EC(6)
    nodes = tree_bfs(tree)
    leaf_ctx, n = {}, 0
    for nd in nodes:
        if nd[0] == "split":
            ec.add(1, nd[1] + 1); ec.add(0, pack_signed(nd[2]))
        else:
            ec.add(1, 0); ec.add(2, nd[1]); ec.add(3, pack_signed(nd[2])); ec.add(4, 0); ec.add(5, 0)       # multiplier (0 + 1) << 0
            leaf_ctx[id(nd)] = n; n += 1
    ec.write_header(bw)
    ec.write_symbols(bw)
    return leaf_ctx, n


def tree_leaf(tree, props):
    nd = tree
    while nd[0] == "split":
        nd = nd[3] if props[nd[1]] > nd[2] else nd[4]
    return nd


def predict(pred, W, N, NW):
    if pred == 0: return 0
    if pred == 1: return W
    if pred == 2: return N
    if pred == 5:
        lo, hi = min(N, W), max(N, W)
        return min(max(N + W - NW, lo), hi)
    raise ValueError(pred)


def write_modular(bw, channels, tree, stream_id):
    """one Modular sub-bitstream (H.2 GroupHeader, its own MA tree, the channels' residuals); trees may test properties 0..3 (channel, stream, y, x)"""
    bw.bool(0)                                             # use_global_tree
    bw.bool(1)                                             # default weighted-predictor parameters
    bw.w(2, 0)                                             # nb_transforms = 0
    leaf_ctx, num_leaves = write_tree(bw, tree)
    ec = EC(num_leaves)
    for ci, ch in enumerate(channels):
        ch = np.asarray(ch, dtype=np.int64)
        if ch.size == 0:
            continue
        h, w = ch.shape
        for y in range(h):
            for x in range(w):
                W = int(ch[y, x - 1]) if x > 0 else (int(ch[y - 1, x]) if y > 0 else 0)
                N = int(ch[y - 1, x]) if y > 0 else W
                NW = int(ch[y - 1, x - 1]) if (x > 0 and y > 0) else W
                leaf = tree_leaf(tree, (ci, stream_id, y, x))
                ec.add(leaf_ctx[id(leaf)], pack_signed(int(ch[y, x]) - leaf[2] - predict(leaf[1], W, N, NW)))
    ec.write_header(bw)
    ec.write_symbols(bw)


def write_splines(bw, splines, quant_adjust=0):
    """K.4: splines = [dict(points=[(x, y), ...] (integers), color=[[32 ints] x 3] (X, Y, B DCT coefficients), sigma=[32 ints])]"""
    ec = EC(6)
    ec.add(2, len(splines) - 1)
    lx = ly = 0
    for i, s in enumerate(splines):
        x, y = s["points"][0]
        if i == 0:
            ec.add(1, x); ec.add(1, y)
        else:
            ec.add(1, pack_signed(x - lx)); ec.add(1, pack_signed(y - ly))
        lx, ly = x, y
    ec.add(0, pack_signed(quant_adjust))
    for s in splines:
        pts = s["points"]
        ec.add(3, len(pts) - 1)
        cx, cy = pts[0]; dx = dy = 0
        for (px, py) in pts[1:]:                            # double-delta coding of the control points
            ndx, ndy = px - cx, py - cy
            ec.add(4, pack_signed(ndx - dx)); ec.add(4, pack_signed(ndy - dy))
            dx, dy, cx, cy = ndx, ndy, px, py
        for c in range(3):
            for v in s["color"][c]:
                ec.add(5, pack_signed(int(v)))
        for v in s["sigma"]:
            ec.add(5, pack_signed(int(v)))
    ec.write_header(bw)
    ec.write_symbols(bw)


def natural_order_len(strategy):
    return COVERED_X[strategy] * COVERED_Y[strategy] * 64


def write_size_header(bw, width, height):
    """SizeHeader (A.2): the small form for multiples of 8 up to 256, else explicit sizes (ratio 0)."""
    if width % 8 == 0 and height % 8 == 0 and width <= 256 and height <= 256:
        bw.bool(1); bw.w(5, height // 8 - 1); bw.w(3, 0); bw.w(5, width // 8 - 1)
    else:
        enc = [(9, 1), (13, 1), (18, 1), (30, 1)]
        bw.bool(0); bw.u32(height, enc); bw.w(3, 0); bw.u32(width, enc)


def write_vardct(width, height, blocks, lf, xfromy=None, bfromy=None, sharpness=None, global_scale=32768, quant_lf=64, gab=True, epf_iters=2,
                 splines=None, spline_quant_adjust=0, x_qm=3, b_qm=2, upsampling=1, up_weights=None, preview=None, as_frame=False, dequant=None, pass_shifts=None):
    """upsampling: 1 / 2 / 4 / 8 — the image is width * upsampling x height * upsampling, the frame is coded at width x height;
    up_weights: {2: [15 floats], 4: [55], 8: [210]} custom upsampling weights in the image metadata (upper triangles of the symmetric kernel matrices);
    preview: the bytes of a frame (write_vardct(..., as_frame=True)) of pw x ph = preview[1], preview[2] pixels, put in front of the image's frames as its preview;
    as_frame: return (frame bytes, width, height) without the file header;
    dequant: {quant table index: (mode, parameters)} — DequantMatrices encodings (I.2.4), values as STORED (the decoder multiplies most of them by 64):
             1 (table 1, IDENTITY): [3][3]; 2 (table 2, DCT2X2): [3][6]; 3 (table 3, DCT4X4): ([3][2] multipliers, bands [3][nb]); 4 (table 9, DCT4X8 / DCT8X4):
             ([3] multipliers, bands); 5 (table 10, AFV): ([3][9], bands of the 4 x 8 part, bands of the 4 x 4 part); 6 (any DCT table): bands [3][nb];
    pass_shifts: [shift of pass 0, ..., shift of pass N - 2] (0..3 each) makes a frame of N = len + 1 passes (2..11): every block then carries
             "coef_passes" = [coef dict of pass 0, ..., of pass N - 1] instead of "coef"; the decoder adds value << shift of every pass (the last pass's shift is 0).
             Such a frame has its LfGlobal / LfGroup / HfGlobal / PassGroup sections apart (TOC of 3 + N entries)."""
    """A one-group VarDCT image.
    blocks: [dict(bx, by, strategy, qf (1..256), coef={channel (0 X, 1 Y, 2 B): {scan position k >= covered cells: quantised value}})] tiling the cell
            grid exactly; lf: int array [3][yb][xb] (X, Y, B quantised LF samples); xfromy / bfromy: int8-range arrays of [ceil(yb/8)][ceil(xb/8)]."""
    assert width <= 256 and height <= 256 and width % 8 == 0 and height % 8 == 0
    xb, yb = width // 8, height // 8
    tw, th = (xb + 7) // 8, (yb + 7) // 8
    occ = np.zeros((yb, xb), bool)
    order = sorted(blocks, key=lambda b: (b["by"], b["bx"]))
    for b in order:                                         # raster order of the first cells = the order the decoder places them in
        cx, cy = COVERED_X[b["strategy"]], COVERED_Y[b["strategy"]]
        assert b["bx"] + cx <= xb and b["by"] + cy <= yb and not occ[b["by"]:b["by"] + cy, b["bx"]:b["bx"] + cx].any(), b
        ys, xs = np.nonzero(~occ)
        assert (ys[0], xs[0]) == (b["by"], b["bx"]), ("blocks must fill the first free cell in raster order", b["bx"], b["by"], xs[0], ys[0])
        occ[b["by"]:b["by"] + cy, b["bx"]:b["bx"] + cx] = True
    assert occ.all()
    bw = BW()
    if not as_frame:
        bw.w(16, 0x0AFF)
        write_size_header(bw, width * upsampling, height * upsampling)
        if preview is None:
            bw.bool(1)                                      # ImageMetadata all_default
        else:
            bw.bool(0); bw.bool(1)                          # not all_default, extra_fields
            bw.w(3, 0)                                      # orientation 1
            bw.bool(0)                                      # no intrinsic size
            bw.bool(1)                                      # have_preview: PreviewHeader (A.3)
            pw, ph = preview[1], preview[2]
            assert pw % 8 == 0 and ph % 8 == 0 and pw <= 256 and ph <= 256
            bw.bool(1)                                      # div8
            bw.u32(ph // 8, [(-1, 16), (-1, 32), (5, 1), (9, 33)]); bw.w(3, 0); bw.u32(pw // 8, [(-1, 16), (-1, 32), (5, 1), (9, 33)])
            bw.bool(0)                                      # no animation
            bw.bool(0); bw.w(2, 0)                          # BitDepth: integer samples, 8 bits
            bw.bool(1)                                      # modular_16bit_buffers
            bw.w(2, 0)                                      # no extra channels
            bw.bool(1)                                      # xyb_encoded
            bw.bool(1)                                      # ColourEncoding all_default (sRGB)
            bw.bool(1)                                      # ToneMapping all_default
            bw.u64(0)                                       # extensions
        if up_weights is None:
            bw.bool(1)                                      # default_m
        else:
            bw.bool(0)
            bw.bool(1)                                      # default opsin inverse matrix
            bw.w(3, (1 if 2 in up_weights else 0) | (2 if 4 in up_weights else 0) | (4 if 8 in up_weights else 0))
            for fac, cnt in ((2, 15), (4, 55), (8, 210)):
                if fac in up_weights:
                    assert len(up_weights[fac]) == cnt
                    for x in up_weights[fac]:
                        bw.f16(float(x))
        if preview is not None:
            bw.align()
            for byte in preview[0]:
                bw.w(8, byte)
        bw.align()
    # FrameHeader
    flags = 16 if splines else 0
    npass = 1 + len(pass_shifts) if pass_shifts else 1
    assert 1 <= npass <= 11
    if flags == 0 and gab and epf_iters == 2 and x_qm == 3 and b_qm == 2 and upsampling == 1 and npass == 1:
        bw.bool(1)
    else:
        bw.bool(0)
        bw.w(2, 0); bw.w(1, 0)                              # regular frame, VarDCT
        bw.u64(flags)
        bw.w(2, {1: 0, 2: 1, 4: 2, 8: 3}[upsampling])       # upsampling
        bw.w(3, x_qm); bw.w(3, b_qm)
        if npass == 1:
            bw.w(2, 0)                                      # one pass
        else:
            bw.u32(npass, [(-1, 1), (-1, 2), (-1, 3), (3, 4)])
            bw.w(2, 0)                                      # num_downsample 0
            for sh in pass_shifts:
                bw.w(2, sh)
        bw.bool(0)                                          # no crop
        bw.w(2, 0)                                          # blend mode: replace (full frame: no source)
        bw.bool(1)                                          # is_last
        bw.w(2, 0)                                          # name length 0
        if gab and epf_iters == 2:
            bw.bool(1)                                      # default restoration filter
        else:
            bw.bool(0); bw.bool(gab)
            if gab:
                bw.bool(0)                                  # default Gaborish weights
            bw.w(2, epf_iters)
            if epf_iters:
                bw.bool(0); bw.bool(0); bw.bool(0)          # default sharpness LUT / channel scales / pass parameters
            bw.u64(0)                                       # loop-filter extensions
        bw.u64(0)                                           # frame extensions
    # ---- the single section (one pass), or LfGlobal / LfGroup / HfGlobal / one PassGroup section per pass
    sec = BW()
    sec_lfgroup = sec if npass == 1 else BW()
    sec_hfglobal = sec if npass == 1 else BW()
    if splines:
        write_splines(sec, splines, spline_quant_adjust)
    sec.bool(1)                                             # default LF dequantisation
    sec.u32(global_scale, [(11, 1), (11, 2049), (12, 4097), (16, 8193)])
    sec.u32(quant_lf, [(-1, 16), (5, 1), (8, 1), (16, 1)])
    sec.bool(1)                                             # default block-context map
    sec.bool(1)                                             # default colour correlation
    sec.bool(0)                                             # no global MA tree
    # LfGroup: LF coefficients (channels Y, X, B), then HF metadata
    sec_lfgroup.w(2, 0)                                             # extra_precision
    lf = np.asarray(lf, dtype=np.int64)
    grad = ("leaf", 5, 0)
    write_modular(sec_lfgroup, [lf[1], lf[0], lf[2]], grad, 1)
    nblk = len(order)
    sec_lfgroup.w(max(0, (xb * yb - 1).bit_length()), nblk - 1)
    xf = np.zeros((th, tw), np.int64) if xfromy is None else np.asarray(xfromy, np.int64)
    bf = np.zeros((th, tw), np.int64) if bfromy is None else np.asarray(bfromy, np.int64)
    info = np.array([[b["strategy"] for b in order], [b["qf"] - 1 for b in order]], np.int64)
    sh = np.zeros((yb, xb), np.int64) if sharpness is None else np.asarray(sharpness, np.int64)
    west = ("leaf", 1, 0)
    write_modular(sec_lfgroup, [xf, bf, info, sh], west, 3)
    # HfGlobal
    if not dequant:
        sec_hfglobal.bool(1)                                         # default dequant matrices
    else:
        def bands(b):
            nb = len(b[0]); assert all(len(r) == nb for r in b) and 1 <= nb <= 16
            sec_hfglobal.w(4, nb - 1)
            for c in range(3):
                for v in b[c]:
                    sec_hfglobal.f16(v)
        sec_hfglobal.bool(0)
        for t in range(17):
            mode, prm = dequant.get(t, (0, None))
            sec_hfglobal.w(3, mode)
            if mode == 1 or mode == 2:
                for c in range(3):
                    for v in prm[c]:
                        sec_hfglobal.f16(v)
            elif mode == 3:
                for c in range(3):
                    for v in prm[0][c]:
                        sec_hfglobal.f16(v)
                bands(prm[1])
            elif mode == 4:
                for c in range(3):
                    sec_hfglobal.f16(prm[0][c])
                bands(prm[1])
            elif mode == 5:
                for c in range(3):
                    for v in prm[0][c]:
                        sec_hfglobal.f16(v)
                bands(prm[1]); bands(prm[2])
            elif mode == 6:
                bands(prm)
            else:
                assert mode == 0
    sec_hfglobal.w(2, 2)                                             # used_orders = 0 (selector 2): natural coefficient orders      (num_presets: 0 bits for one group)
    # PassGroup: per varblock (raster order of first cells), channels Y, X, B: nonzero count, then the coefficients up to the last nonzero one
    pass_secs = []
    for p in range(npass):
        hf = EC(495 * 15)
        for b in order:
            size, covered = natural_order_len(b["strategy"]), COVERED_X[b["strategy"]] * COVERED_Y[b["strategy"]]
            src = b.get("coef", {}) if npass == 1 else b["coef_passes"][p]
            for c in (1, 0, 2):
                co = {int(k): int(v) for k, v in src.get(c, {}).items() if v}
                assert all(covered <= k < size for k in co), (b["strategy"], sorted(co)[:4], covered, size)
                hf.add(0, len(co))
                if co:
                    for k in range(covered, max(co) + 1):
                        hf.add(0, pack_signed(co.get(k, 0)))
        if p > 0:
            sec_hfglobal.w(2, 2)                            # used_orders = 0 for this pass too
        hf.write_header(sec_hfglobal)                       # HfPass: the pass's histograms
        ps = sec if npass == 1 else BW()
        hf.write_symbols(ps)
        pass_secs.append(ps)
    parts = [sec.bytes()] if npass == 1 else [sec.bytes(), sec_lfgroup.bytes(), sec_hfglobal.bytes()] + [ps.bytes() for ps in pass_secs]
    body = b"".join(parts)
    # TOC
    bw.bool(0)                                              # not permuted
    bw.align()
    for part in parts:
        bw.u32(len(part), [(10, 0), (14, 1024), (22, 17408), (30, 4211712)])
    bw.align()
    if as_frame:
        return bw.bytes() + body, width, height
    return bw.bytes() + body


if __name__ == "__main__":
    import os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jxl_ref
    yy, xx = np.mgrid[0:32, 0:32]
    lf = np.stack([np.zeros((32, 32), np.int64), 6000 + 40 * xx + 25 * yy, 5000 + 10 * xx])
    data = write_vardct(256, 256, [dict(bx=0, by=0, strategy=24, qf=8, coef={1: {1024: 5, 1030: -3, 4000: 2}, 0: {1100: 1}, 2: {1500: -2}})], lf)
    px, info, _ = jxl_ref.decode(data)
    print(len(data), px.shape, px[..., :3].min(), px[..., :3].max(), px[::64, ::64, 1])
