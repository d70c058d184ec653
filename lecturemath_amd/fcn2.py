"""Recipes of the second FCN engine (csrc/lm_fcn2.hip): for every layer of FCN-LectureNet (AccessMath/lecturenet_v1/FCN_lecturenet.py
:260-323, :364-403) the K walk of the gather-GEMM kernel -- which (plane, tap) pair feeds each of the four k-groups of every
32-deep slice -- and the weights packed as that kernel's A fragments.  numpy only rearranges weights once; all arithmetic of a
forward pass runs in liblecturemath_hip.so.

Vocabulary (see the header of lm_fcn2.hip): a TENSOR is C/8 planes of 16-byte slots (8 channels of a pixel as f16, hi parts, then
optionally the lo parts); a layer stages its input a CHUNK of planes at a time into LDS; a PAIR is (plane of the chunk, tap dy, dx):
the 8 k-values one lane group reads with one ds_read_b128; a SLICE is four pairs = one 16x16x32 MFMA step; consecutive slices are
fetched as weight GROUPS.  A pair plane (network input, diff) holds {c0 c1 c2 0 | the same of pixel x + 1} per slot.
"""
import numpy as np

EPI_PO, EPI_T, EPI_TC, EPI_TC2, EPI_V = 0, 1, 2, 3, 4      # EPI_V: a head's row convolution fused with its vertical sum
T_X0P, T_PRE0, T_POOL0, T_MID, T_UPT0, T_CU0, T_XUP, T_DP, T_P1, T_P2, N_TENSORS = 0, 1, 6, 11, 12, 17, 21, 22, 23, 24, 25
LDS_TWO_WORKGROUPS = 80 * 1024        # a workgroup's LDS for two of them to share a CU's 160 KB


def geom(kh, kw, terms, nc=1):
    pw, ph = 16 * nc + kw - 1, 16 + kh - 1
    pls = (ph * pw * 16 + 255) & ~255
    return pw, ph, pls, (2 if terms in (2, 3) else 1)


# operand formats (the TERMS parameter of lm_k_g2): MFMA products per operand pair
FORMAT_PRODUCTS = {1: 1, 2: 2, 3: 3, 4: 2}
FORMAT_NAMES = {"f16": 1, "a2": 2, "f16x3": 3, "w2": 4}     # a2: activations hi + lo; w2: weights hi + lo


def octet_pair(plane, dy, dx, cbase):
    """k-values of an ordinary plane at tap (dy, dx): channels cbase .. cbase + 7 of the weights' input axis"""
    return {"plane": plane, "dy": dy, "dx": dx, "kmap": [(cbase + j, dy, dx) for j in range(8)]}


def pairplane_pair(plane, dy, dx, cbase, kw_total):
    """k-values of a pair plane read at tap (dy, dx): three channels of that tap, a zero, three channels of tap dx + 1, a zero"""
    km = [(cbase + j, dy, dx) for j in range(3)] + [None]
    km += [((cbase + j, dy, dx + 1) if dx + 1 < kw_total else None) for j in range(3)] + [None]
    return {"plane": plane, "dy": dy, "dx": dx, "kmap": km}


def row_channel(mt, m, r, epi):
    """channel (relative to the workgroup's first) held by row r of channel tile m -- lm_k_g2's epilogues"""
    if epi not in (EPI_T, EPI_V) and m < 2 * (mt // 2):
        return 32 * (m // 2) + 8 * (r >> 2) + 4 * (m & 1) + (r & 3)
    return 16 * m + r


def have_instance(kh, kw, terms, mt, epi, nc=1, loader=0):
    """the kernel instances csrc/lm_fcn2.hip compiles (lm_g2_launch)"""
    v = (nc, loader)
    if (kh, kw) == (3, 3) and epi == EPI_PO:
        if v == (1, 0):
            return terms in (1, 3, 4) and 1 <= mt <= 4
        if v == (1, 1):
            return (terms == 1 and 1 <= mt <= 4) or (terms == 4 and mt in (1, 2))
        return v == (2, 0) and terms in (1, 3, 4) and 1 <= mt <= (3 if terms == 1 else 2)
    if (kh, kw) == (1, 1) and epi in (EPI_TC, EPI_TC2):
        return v == (1, 0) and terms in (1, 3) and 1 <= mt <= 4
    if (kh, kw) == (1, 7) and epi == EPI_T:
        return v in ((1, 0), (2, 0)) and 1 <= terms <= 4 and mt == 1
    if (kh, kw) == (1, 7) and epi == EPI_V:
        return v in ((1, 0), (2, 0)) and terms in (3, 4) and mt == 1
    if (kh, kw) == (7, 7) and epi == EPI_PO:
        return v in ((1, 0), (1, 1), (2, 0)) and 1 <= terms <= 4 and mt in (1, 2)
    return False


def table_bytes(nslices):
    """LDS bytes of the slice table: 16 bytes per slice (the four k-groups' patch offsets) + two entries of look-ahead"""
    return ((nslices + 2) * 16 + 255) & ~255


def lds_bytes(kh, kw, terms, npc, nslices, pdouble, wbuf, ngroups, nc=1):
    _, _, pls, nhl = geom(kh, kw, terms, nc)
    return table_bytes(nslices) + (2 if pdouble else 1) * npc * nhl * pls + (2 if ngroups > 1 else 1) * wbuf


def build(w_list, chunks, kh, kw, terms, mt, epi, gsize=None, pdouble=None, lds_target=LDS_TWO_WORKGROUPS, nc=1, loader=0):
    """w_list: one [cout][cin][KH][KW] float32 array (or four, one per (dy, dx) of a transposed convolution).
    chunks: [{"planes": [(tensor, octet), ...], "pairs": [pair, ...]}], every chunk with the same number of planes.
    Returns (desc int32 array for lm_fcn2_set_layer, packed weights as bytes array, wblocks)."""
    if epi == EPI_V and not have_instance(kh, kw, terms, mt, epi, nc, loader):       # no fused instance for this format: rows + vertical-sum kernel
        epi = EPI_T
    if not have_instance(kh, kw, terms, mt, epi, nc, loader):       # a variant the library does not hold: the plain one
        nc, loader = 1, 0
    pw, ph, pls, nhl = geom(kh, kw, terms, nc)
    nwl = 2 if terms >= 3 else 1
    cout = w_list[0].shape[0]
    assert cout % (16 * mt) == 0, (cout, mt)
    npc = len(chunks[0]["planes"])
    slices, kidx = [], []                        # slices: (four LDS offsets, chunk)
    wshape = w_list[0].shape
    for ci, ch in enumerate(chunks):
        assert len(ch["planes"]) == npc
        prs = ch["pairs"]
        for s0 in range(0, len(prs), 4):
            four = list(prs[s0:s0 + 4])
            while len(four) < 4:                 # unused k-groups: zero weights over data that is certainly finite (the slice's first pair)
                four.append({"plane": four[0]["plane"], "dy": four[0]["dy"], "dx": four[0]["dx"], "kmap": [None] * 8})
            offs = [p["plane"] * nhl * pls + (p["dy"] * pw + p["dx"]) * 16 for p in four]
            assert all(0 <= p["dy"] < kh and 0 <= p["dx"] < kw and 0 <= p["plane"] < npc for p in four)
            slices.append((offs, ci))
            idx = np.full(32, -1, np.int64)
            for g, p in enumerate(four):
                for j, km in enumerate(p["kmap"]):
                    if km is not None:
                        idx[g * 8 + j] = (km[0] * wshape[2] + km[1]) * wshape[3] + km[2]
            kidx.append(idx)
    nslices = len(slices)
    kidx = np.stack(kidx)                                           # [nslices][32]
    # weight groups: consecutive slices of one chunk
    per_slice = mt * nwl * 1024
    if pdouble is None:
        pdouble = len(chunks) > 1
    if gsize is None:
        gsize = 1
        for cand in range(2, 17):
            # two weight buffers within the target for the layers that share a CU's LDS between two workgroups (the run-time ring takes a
            # third when it fits 80 KB); three where a workgroup has the CU to itself
            if lds_bytes(kh, kw, terms, npc, nslices, pdouble, cand * per_slice * (3 if lds_target > LDS_TWO_WORKGROUPS else 2) // 2, 2, nc) <= lds_target:
                gsize = cand
    groups = []
    s = 0
    while s < nslices:                       # per chunk: as few groups as gsize allows, sizes as even as possible
        n = 1
        while s + n < nslices and slices[s + n][1] == slices[s][1]:
            n += 1
        k = (n + gsize - 1) // gsize
        for i in range(k):
            cnt = n // k + (1 if i < n % k else 0)
            groups.append([s, cnt, slices[s][1]])
            s += cnt
    wbuf = max(g[1] for g in groups) * per_slice
    # weights: [parity][block][slice][tile][hi|lo][lane = kgroup * 16 + row][8]
    nblocks = cout // (16 * mt)
    rows = np.array([[[b * 16 * mt + row_channel(mt, m, r, epi) for r in range(16)] for m in range(mt)] for b in range(nblocks)])     # [b][m][r]
    packed = []
    for w in w_list:
        assert w.shape == wshape
        w2 = np.concatenate([w.reshape(cout, -1).astype(np.float32), np.zeros((cout, 1), np.float32)], axis=1)
        ws = w2[:, kidx]                                            # [cout][nslices][32]   (index -1 = the zero column)
        a = ws[rows]                                                # [b][m][r][nslices][32]
        a = a.reshape(nblocks, mt, 16, nslices, 4, 8).transpose(0, 3, 1, 4, 2, 5)       # b, slice, m, kgroup, r, j
        hi = a.astype(np.float16)
        parts = [hi]
        if nwl == 2:
            parts.append((a - hi.astype(np.float32)).astype(np.float16))
        packed.append(np.stack(parts, axis=3).reshape(nblocks, nslices, mt, nwl, 64, 8))
    wpk = np.ascontiguousarray(np.stack(packed))                    # [parity][block]...
    planes = [v for ch in chunks for pl in ch["planes"] for v in pl]
    # flags: bits 0-3 column tiles, bit 8 loader wave, bits 16.. the LDS target in KB (the run-time weight ring stays inside it)
    desc = [kh, kw, terms, mt, epi, len(chunks), npc, len(groups), nslices, nc | (loader << 8) | ((lds_target // 1024) << 16), 1 if pdouble else 0, wbuf, cout]
    desc += planes + [v for g in groups for v in g] + [v for s_ in slices for v in s_[0]]
    need = lds_bytes(kh, kw, terms, npc, nslices, pdouble, wbuf, len(groups), nc)
    return np.asarray(desc, np.int32), wpk, len(w_list) * nblocks, need


def conv_chunks(inputs, kh, kw, co):
    """ordinary planes: inputs = [(tensor, octets)] concatenated along the weights' input axis; chunks of `co` planes, pairs in
    (tap, plane) order"""
    planes, cbase = [], 0
    for t, n in inputs:
        for o in range(n):
            planes.append((t, o, cbase))
            cbase += 8
    assert len(planes) % co == 0
    chunks = []
    for c0 in range(0, len(planes), co):
        pl = planes[c0:c0 + co]
        pairs = [octet_pair(i, dy, dx, p[2]) for dy in range(kh) for dx in range(kw) for i, p in enumerate(pl)]
        chunks.append({"planes": [(p[0], p[1]) for p in pl], "pairs": pairs})
    return chunks


def pick_mt(cout, tiles, prefer=(4, 3, 2)):
    """channel tiles per workgroup: the largest that divides cout / 16 and still leaves the grid >= 1.5 workgroups per CU"""
    nt = cout // 16
    ok = [m for m in prefer if nt % m == 0] or [1]
    for m in ok:
        if tiles * (nt // m) >= 384:
            return m
    return ok[-1]


def conv_layer(w, inputs, terms, tiles, mt=None, nc=1, loader=0, lds_target=LDS_TWO_WORKGROUPS, gsize=None):
    """tiles: 16 x 16 pixel tiles of the layer's grid (a 16 x 32 tile counts as two)"""
    cout, _, kh, kw = w.shape
    if not have_instance(kh, kw, terms, mt or pick_mt(cout, tiles // nc), EPI_PO, nc, loader):
        nc, loader = 1, 0
    mt = mt or pick_mt(cout, tiles // nc)
    total = sum(n for _, n in inputs)
    best = None
    for co in (4, 3, 2, 1):
        if total % co:
            continue
        chunks = conv_chunks(inputs, kh, kw, co)
        pd = len(chunks) > 1
        r = build([w], chunks, kh, kw, terms, mt, EPI_PO, pdouble=pd, nc=nc, loader=loader, lds_target=lds_target, gsize=gsize)
        key = (r[3] > lds_target, int(r[0][8]))      # room for two workgroups per CU first, then the fewest slices
        if best is None or key < best[0]:
            best = (key, r)
    return best[1]


def pixel_chunks(feat_tensor, nf, dp_tensor, kh_total, kw_total, octets=2):
    """(diff, features) input of the 7x7 pixel branch / the output row convolution: chunks of two feature octets + the diff pair plane;
    the pair plane's taps (kernel columns 0, 2, 4, 6: a slot covers two) are spread over the chunks' last slices.
    Weight input axis: 0..2 diff, 3.. features (cat((diff, features)), FCN_lecturenet.py:383-395).
    octets = 1: chunks of ONE feature octet + the pair plane (half the patch bytes in LDS: a 16 x 32 tile of a split format fits)."""
    if octets == 1:
        return pixel_chunks_1(feat_tensor, nf, dp_tensor, kh_total, kw_total)
    assert nf % 2 == 0
    nch = nf // 2
    dp_taps = list(range(0, kw_total, 2))
    share = [dp_taps[i * len(dp_taps) // nch:(i + 1) * len(dp_taps) // nch] for i in range(nch)]
    chunks = []
    for c in range(nch):
        pairs = []
        for dy in range(kh_total):
            row = []
            for dx in range(0, kw_total - 1, 2):
                row += [octet_pair(0, dy, dx, 3 + 16 * c), octet_pair(1, dy, dx, 3 + 16 * c + 8),
                        octet_pair(0, dy, dx + 1, 3 + 16 * c), octet_pair(1, dy, dx + 1, 3 + 16 * c + 8)]
            if kw_total & 1:
                row += [octet_pair(0, dy, kw_total - 1, 3 + 16 * c), octet_pair(1, dy, kw_total - 1, 3 + 16 * c + 8)]
            row += [pairplane_pair(2, dy, dx, 0, kw_total) for dx in share[c]]
            pairs += row
        chunks.append({"planes": [(feat_tensor, 2 * c), (feat_tensor, 2 * c + 1), (dp_tensor, 0)], "pairs": pairs})
    return chunks


def pixel_chunks_1(feat_tensor, nf, dp_tensor, kh_total, kw_total):
    """one feature octet per chunk: per kernel row its kw taps, then the chunk's share of the pair plane's taps"""
    dp_taps = list(range(0, kw_total, 2))
    share = [dp_taps[i * len(dp_taps) // nf:(i + 1) * len(dp_taps) // nf] for i in range(nf)]
    chunks = []
    for c in range(nf):
        pairs = []
        for dy in range(kh_total):
            pairs += [octet_pair(0, dy, dx, 3 + 8 * c) for dx in range(kw_total)]
            pairs += [pairplane_pair(1, dy, dx, 0, kw_total) for dx in share[c]]
        chunks.append({"planes": [(feat_tensor, c), (dp_tensor, 0)], "pairs": pairs})
    return chunks


def text_rec_rows(w_text, w_rec):
    """text mask (7x7, 1 output) + reconstruction (3x3, 3 outputs) as one 1x7 row convolution with 16 outputs: 0..6 text kernel rows,
    7 + kh * 3 + co reconstruction rows with their three taps centred (lm_k_vsum2_text_rec adds the rows up)"""
    cin = w_text.shape[1]
    rows = np.zeros((16, cin, 1, 7), np.float32)
    rows[0:7, :, 0, :] = w_text[0].transpose(1, 0, 2)
    rows[7:16, :, 0, 2:5] = w_rec.transpose(2, 0, 1, 3).reshape(9, cin, 3)
    return rows


def out_rows(w_out):
    """7x7 convolution with one output as a 1x7 row convolution with 7 (of 16) outputs: row kh = kernel row kh"""
    _, cin, k, _ = w_out.shape
    rows = np.zeros((16, cin, 1, k), np.float32)
    rows[0:k, :, 0, :] = w_out[0].transpose(1, 0, 2)
    return rows
