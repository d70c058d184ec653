"""Host side of the HIP FCN-LectureNet: BatchNorm folding, weight packing into MFMA fragment order, and a class with
the reference's inference API (AccessMath/lecturenet_v1/FCN_lecturenet.py: CreateFromConfig :620-659, load_state_dict,
eval, cuda, binarize :430-505) so the step-01 worker and test_FCN_binarizer.py can use it unchanged.

All convolution arithmetic runs in liblecturemath_hip.so (lm_fcn.hip); numpy here only rearranges weights once.
"""
import ctypes
import os

import numpy as np

from . import _lib
from .device import Backend

BN_EPS = 1e-5

# layer ids of lm_fcn.hip
L_DOWN, L_MID, L_UPT, L_UPC, L_TEXT, L_REC, L_PX1, L_PX2, L_OUT = 0, 5, 6, 11, 16, 17, 18, 19, 20

WIDTH_KEYS = [  # CreateFromConfig :621-646, with its defaults
    ("FCN_BINARIZER_NET_DOWN_CONV_FILTERS_1", 16), ("FCN_BINARIZER_NET_DOWN_CONV_FILTERS_2", 32),
    ("FCN_BINARIZER_NET_DOWN_CONV_FILTERS_3", 64), ("FCN_BINARIZER_NET_DOWN_CONV_FILTERS_4", 128),
    ("FCN_BINARIZER_NET_DOWN_CONV_FILTERS_5", 256), ("FCN_BINARIZER_NET_MIDDLE_CONV_FILTERS_MIDDLE", 512),
    ("FCN_BINARIZER_NET_UPSAMPLE_FILTERS_5", 256), ("FCN_BINARIZER_NET_UP_CONV_FILTERS_5", 256),
    ("FCN_BINARIZER_NET_UPSAMPLE_FILTERS_4", 128), ("FCN_BINARIZER_NET_UP_CONV_FILTERS_4", 128),
    ("FCN_BINARIZER_NET_UPSAMPLE_FILTERS_3", 64), ("FCN_BINARIZER_NET_UP_CONV_FILTERS_3", 64),
    ("FCN_BINARIZER_NET_UPSAMPLE_FILTERS_2", 32), ("FCN_BINARIZER_NET_UP_CONV_FILTERS_2", 32),
    ("FCN_BINARIZER_NET_UPSAMPLE_FILTERS_1", 16), ("FCN_BINARIZER_NET_UP_CONV_FILTERS_1", 16),
    ("FCN_BINARIZER_NET_PIXEL_FEATURES_1", 32), ("FCN_BINARIZER_NET_PIXEL_FEATURES_2", 16),
]


def _np(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)


def fold_bn(w, b, sd, bn, out_axis):
    """conv (or transposed conv) followed by eval-mode BatchNorm -> one affine conv, in fp32."""
    g, beta = _np(sd[bn + ".weight"]).astype(np.float32), _np(sd[bn + ".bias"]).astype(np.float32)
    mean, var = _np(sd[bn + ".running_mean"]).astype(np.float32), _np(sd[bn + ".running_var"]).astype(np.float32)
    s = (g / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32)
    shape = [1] * w.ndim
    shape[out_axis] = -1
    return (w * s.reshape(shape)).astype(np.float32), ((b - mean) * s + beta).astype(np.float32)


def pack_mfma(w_oikk, ck, cin_map, cin_padded):
    """[Cout][Cin][K][K] -> [chunk][tap][kstep][nblock][lane 64][4] for v_mfma_f32_32x32x2_f32:
    element e of lane l = W[co = nblock*32 + (l & 31)][ci = chunk*ck + kstep*8 + 4*(l >> 5) + e][tap].
    cin_map[i] = position of logical input channel i in the padded channel space of the input buffer(s)."""
    cout, cin, k, _ = w_oikk.shape
    nblocks = (cout + 31) // 32
    wp = np.zeros((nblocks * 32, cin_padded, k * k), np.float32)
    wp[:cout][:, np.asarray(cin_map)] = w_oikk.reshape(cout, cin, k * k)
    nchunks, ks = cin_padded // ck, ck // 8
    # wp[co, ci, tap] -> [chunk, ks, half, e, tap, nblock, j]
    a = wp.reshape(nblocks, 32, nchunks, ks, 2, 4, k * k)            # nb, j, chunk, ks, half, e, tap
    a = a.transpose(2, 6, 3, 0, 4, 1, 5)                              # chunk, tap, ks, nb, half, j, e
    return np.ascontiguousarray(a).reshape(-1)


def pack_mfma_h(w_oikk, cin_map, cin_logical):
    """fp16-split packing for lm_k_conv_mfma_h: [chunk][tap][nblock][hi|lo][lane 64][8 halfs], element j of lane l =
    W[co = nblock*32 + (l & 31)][ci = chunk*16 + 8*(l >> 5) + j][tap]; hi = f16(w), lo = f16(w - hi).
    cin_map[i] = position of weight input channel i among the cin_logical concatenated input channels (padded to 16 here).
    Returned as a float32 view (two halfs per float) for lm_fcn_set_layer."""
    cout, cin, kh, kw = w_oikk.shape                                    # square kernels, or the 1 x K rows of pack_rows_h
    taps = kh * kw
    nblocks = (cout + 31) // 32
    cpad = ((cin_logical + 15) // 16) * 16
    wp = np.zeros((nblocks * 32, cpad, taps), np.float32)
    wp[:cout][:, np.asarray(cin_map)] = w_oikk.reshape(cout, cin, taps)
    hi = wp.astype(np.float16)
    lo = (wp - hi.astype(np.float32)).astype(np.float16)
    both = np.stack([hi, lo])                                           # hl, co, ci, tap
    a = both.reshape(2, nblocks, 32, cpad // 16, 2, 8, taps)           # hl, nb, j, chunk, half, e, tap
    a = a.transpose(3, 6, 1, 0, 4, 2, 5)                                # chunk, tap, nb, hl, half, j, e
    return np.ascontiguousarray(a).reshape(-1).view(np.float32)


def pack_rows_h(w_oikk, cin_map, cin_logical):
    """A K x K convolution with NV <= 3 outputs as a 1 x K row convolution with K * NV outputs (lm_rowconv_layer in lm_fcn.hip):
    virtual output kh * NV + co holds kernel row kh of channel co; lm_k_vsum adds the rows up."""
    nv, cin, k, _ = w_oikk.shape
    rows = np.ascontiguousarray(w_oikk.transpose(2, 0, 1, 3)).reshape(k * nv, cin, 1, k)      # [kh][co][ci][kw] -> [kh * NV + co][ci][1][kw]
    return pack_mfma_h(rows, cin_map, cin_logical)


def pack_text_rec_rows_h(w_text, w_rec, cin_map, cin_logical):
    """Text mask (7x7, 1 output) and reconstruction (3x3, 3 outputs) over the same input as ONE 1 x 7 row convolution with 16
    outputs (lm_text_rec_heads): 0..6 = text kernel rows, 7 + kh * 3 + co = reconstruction rows, their taps centred (kw + 2)."""
    cin = w_text.shape[1]
    rows = np.zeros((16, cin, 1, 7), np.float32)
    rows[0:7, :, 0, :] = w_text[0].transpose(1, 0, 2)                   # [ci][kh][kw] -> [kh][ci][kw]
    rows[7:16, :, 0, 2:5] = w_rec.transpose(2, 0, 1, 3).reshape(9, cin, 3)      # [co][ci][kh][kw] -> [kh * 3 + co][ci][kw]
    return pack_mfma_h(rows, cin_map, cin_logical)


def pack_small(w_oikk, cin_map, cin_padded):
    """[Cout<=4][Cin][K][K] -> [chunk of 8 channels][tap][8] for Cout == 1, [chunk][tap][8][4] otherwise (lm_k_conv_small)."""
    cout, cin, k, _ = w_oikk.shape
    lanes = 1 if cout == 1 else 4
    assert cin_padded % 8 == 0
    out = np.zeros((k * k, cin_padded, lanes), np.float32)
    out[:, np.asarray(cin_map), :cout] = w_oikk.reshape(cout, cin, k * k).transpose(2, 1, 0)
    out = out.reshape(k * k, cin_padded // 8, 8, lanes).transpose(1, 0, 2, 3)
    return np.ascontiguousarray(out).reshape(-1)


def _pad8(c):
    return (c + 7) & ~7


class FcnEngine:
    """Device network built from a reference state_dict."""

    # Operand formats of the "mixed" assignment (fcn2.FORMAT_NAMES).  Below full resolution: plain f16 (all 14 layers together change the
    # logits by 3e-5, profiles/r03_fcn_layer_precision.*).  The six full-resolution layers change them by 0.5-3e-4 EACH on plain f16:
    # round 3 kept all of them on the three-product split (2.6e-5 in all); round 4 measured every assignment by what the path does with
    # the logits -- max |logit - oracle| and BINARY FLIPS against the oracle's binarization, 3 seeds at 1920x1080
    # (profiles/r04_fcn_formats.*) -- and moved conv_up_1, the text / reconstruction heads and conv_pixels_1 to "w2" (weights hi + lo,
    # activations hi: two products, and their input tensors need no lo planes at all): 2.1e-4 (bar 1e-3), flips 392 / 26 / 9 of 2.07 M
    # pixels on random-init logits that crowd the threshold (std 0.06-0.12; all-f16x3: 45 / 2 / 1; all-f16: 1034 / 70 / 47).
    # conv_pixels_2 and conv_out stay on the split: each alone costs 1.5-3e-4 on two products.
    MIXED_FORMATS = {L_DOWN: "f16x3", L_UPC + 4: "w2", L_TEXT: "w2", L_REC: "w2", L_PX1: "w2", L_PX2: "f16x3", L_OUT: "f16x3"}

    # kernel variant per layer, (column tiles per wave, loader wave), measured per layer at 1920x1080 (profiles/r04_variants_*.txt): 16 x 32
    # tiles where the weights are re-fetched per tile at full resolution and the instance keeps two workgroups per CU; the loader wave
    # in the two layers with one workgroup per CU and two channel tiles
    # {layer: channel tiles per workgroup} where fcn2.pick_mt's rule is not the fastest (profiles/r04_mt_{default,a,b}.txt: conv_down_1 on ONE
    # tile = 24,480 small workgroups at 114 VGPRs, 139 -> 124-130 us -- it is a 250 MB store; every other layer is fastest on pick_mt's choice)
    DEFAULT_MT = {L_DOWN: 1}
    DEFAULT_LDS = {}        # {layer: LDS bytes a workgroup may take}; default 80 KB = two workgroups per CU
    DEFAULT_VARIANTS = {L_UPC + 4: (2, 0), L_TEXT: (2, 0), L_PX1: (2, 0), L_PX2: (2, 0), L_MID: (1, 1), L_UPC: (1, 1)}

    def __init__(self, widths, pixel_kernel, kernel, max_h, max_w, lib=None, precision="mixed", formats=None):
        """precision:
        "mixed" (default) -- the planar engine (csrc/lm_fcn2.hip): f16 hi + lo split operands (three MFMAs per product, ~22 bits per
            operand) in the full-resolution layers, plain f16 operands below; needs the shipped kernel sizes (7x7 pixel branch, 3x3
            elsewhere) and widths that are multiples of 16, otherwise this falls back to "f16x3";
        "planar-f16x3" / "planar-f16" -- the planar engine with one format everywhere;
        "f16x3" / "f16x2" / "f16" / "fp32" -- the first engine (csrc/lm_fcn.hip): fp32 activations, operands split while staged
            (three, two or one f16 MFMA per product) or exact fp32 MFMA chains."""
        assert precision in ("mixed", "planar-f16x3", "planar-f16", "f16x3", "f16x2", "f16", "fp32")
        # planar engine only: {layer id: "f16" | "a2" | "w2" | "f16x3"} overriding the precision's assignment (fcn2.FORMAT_NAMES)
        if formats is None and os.environ.get("LM_FCN_FORMATS"):        # experiments: "15=w2,18=a2"
            formats = {int(k): v for k, v in (kv.split("=") for kv in os.environ["LM_FCN_FORMATS"].split(","))}
        self.formats = dict(formats or {})
        # planar engine only: {layer id: (column tiles per wave 1 | 2, loader wave 0 | 1)} -- the kernel variant of a layer (lm_k_g2's NC, LOADER)
        self.variants = dict(self.DEFAULT_VARIANTS)
        if os.environ.get("LM_FCN_VARIANTS"):                           # experiments: "18=2:0,5=1:1"
            for kv in os.environ["LM_FCN_VARIANTS"].split(","):
                k, v = kv.split("=")
                self.variants[int(k)] = tuple(int(x) for x in v.split(":"))
        self.lib = lib or _lib.load()
        self.be = Backend(self.lib)
        self.widths = [int(v) for v in widths]
        self.pk, self.kk = int(pixel_kernel), int(kernel)
        self.max_h, self.max_w = max_h, max_w
        planar_ok = self.pk == 7 and self.kk == 3 and all(v % 16 == 0 for v in self.widths)
        if precision in ("mixed", "planar-f16x3", "planar-f16") and not planar_ok:
            precision = {"mixed": "f16x3", "planar-f16x3": "f16x3", "planar-f16": "f16"}[precision]
        self.precision = precision
        self.planar = precision in ("mixed", "planar-f16x3", "planar-f16")
        self.handle = None
        self.handle2 = None
        if self.planar:
            return          # lm_fcn2_create needs the tensors' lo flags: created by load_state_dict
        arr = (ctypes.c_int32 * 18)(*self.widths)
        self.handle = self.lib.lm_fcn_create(arr, self.pk, self.kk, max_h, max_w)
        if not self.handle:
            raise _lib.LecturemathError(_lib.LM_ERR_ARG, self.lib.last_error())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lm_fcn_destroy(self.handle)
            self.handle = None
        if getattr(self, "handle2", None):
            self.lib.lm_fcn2_destroy(self.handle2)
            self.handle2 = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _set(self, layer, w, b, cin, cout, k, ck):
        w = np.ascontiguousarray(w, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        self.lib.check(self.lib.lm_fcn_set_layer(self.handle, layer, w.ctypes.data, w.size, b.ctypes.data, b.size, cin, cout, k, ck))

    def layer_terms(self, layer):
        """operand format of a layer of the planar engine (the TERMS parameter of lm_k_g2, fcn2.FORMAT_NAMES)"""
        from . import fcn2 as f2
        if layer in self.formats:
            v = self.formats[layer]
            return f2.FORMAT_NAMES[v] if isinstance(v, str) else int(v)
        if self.precision == "planar-f16":
            return 1
        if self.precision == "planar-f16x3":
            return 3
        return f2.FORMAT_NAMES[self.MIXED_FORMATS.get(layer, "f16")]

    def _load_planar(self, sd):
        """recipes of csrc/lm_fcn2.hip (lecturemath_amd/fcn2.py)"""
        from . import fcn2 as f2
        d1, d2, d3, d4, d5, mid, u5, c5, u4, c4, u3, c3, u2, c2, u1, c1, pm1, pm2 = self.widths
        downs = [d1, d2, d3, d4, d5]

        def conv_bn(name):
            w, b = _np(sd[name + ".0.weight"]).astype(np.float32), _np(sd[name + ".0.bias"]).astype(np.float32)
            return fold_bn(w, b, sd, name + ".1", 0)

        def tiles(level):
            return (((self.max_h >> level) + 15) // 16) * (((self.max_w >> level) + 15) // 16)

        T = self.layer_terms

        def V(layer):
            nc, loader = self.variants.get(layer, (1, 0))
            return {"nc": nc, "loader": loader}
        lds_over = {}
        if os.environ.get("LM_FCN2_LDS"):                                  # experiments: "18=53000,15=53000" (bytes of LDS a workgroup may take)
            lds_over = {int(k): int(v) for k, v in (kv.split("=") for kv in os.environ["LM_FCN2_LDS"].split(","))}
        lds_over = {**self.DEFAULT_LDS, **lds_over}

        def LDS(layer, default=f2.LDS_TWO_WORKGROUPS):
            return lds_over.get(layer, default)
        mt_over = {}
        if os.environ.get("LM_FCN2_MT"):                                   # experiments: channel tiles per workgroup, "0=1,10=2"
            mt_over = {int(k): int(v) for k, v in (kv.split("=") for kv in os.environ["LM_FCN2_MT"].split(","))}
        mt_over = {**self.DEFAULT_MT, **mt_over}
        gs_over = {}
        if os.environ.get("LM_FCN2_GSIZE"):                                # experiments: slices per weight group, "18=2,15=3"
            gs_over = {int(k): int(v) for k, v in (kv.split("=") for kv in os.environ["LM_FCN2_GSIZE"].split(","))}
        recipes = {}
        # feature octets per chunk of conv_pixels_2: one, so that the 16 x 32 tile's patch planes of the split format leave room for two
        # workgroups per CU (two octets: 88 KB of LDS, one workgroup, 757 us; one octet: 76 KB, 338 us; 16 x 16 tiles: 364 us)
        px_octets = int(os.environ.get("LM_FCN2_PX_OCTETS", "1" if self.variants.get(L_PX2, (1, 0))[0] == 2 else "2"))
        deep_lds = int(os.environ.get("LM_FCN2_DEEP_LDS", str(f2.LDS_TWO_WORKGROUPS)))    # experiments: LDS of the layers with <= ~1 workgroup per CU

        def lds_for(layer_tiles, cout, mt_guess=4):
            return deep_lds if layer_tiles * max(1, cout // (16 * mt_guess)) <= 320 else f2.LDS_TWO_WORKGROUPS
        # encoder: layer 1 reads the input pair plane (3 channels, two horizontal taps per slot)
        w, b = conv_bn("conv_down_block_1")
        pairs = [f2.pairplane_pair(0, dy, dx, 0, 3) for dy in range(3) for dx in (0, 2)]
        recipes[L_DOWN] = (f2.build([w], [{"planes": [(f2.T_X0P, 0)], "pairs": pairs}], 3, 3, T(L_DOWN), mt_over.get(L_DOWN) or f2.pick_mt(d1, tiles(0) // V(L_DOWN)["nc"]), f2.EPI_PO, lds_target=LDS(L_DOWN), **V(L_DOWN)), b)
        cin = [3] + downs
        for n in range(1, 5):
            w, b = conv_bn("conv_down_block_%d" % (n + 1))
            recipes[L_DOWN + n] = (f2.conv_layer(w, [(f2.T_POOL0 + n - 1, cin[n] // 8)], T(L_DOWN + n), tiles(n), mt=mt_over.get(L_DOWN + n), lds_target=LDS(L_DOWN + n, lds_for(tiles(n), downs[n])), **V(L_DOWN + n)), b)
        w, b = conv_bn("mid_block")
        recipes[L_MID] = (f2.conv_layer(w, [(f2.T_POOL0 + 4, d5 // 8)], T(L_MID), tiles(5), mt=mt_over.get(L_MID), lds_target=LDS(L_MID, lds_for(tiles(5), mid, 2)), **V(L_MID)), b)
        ups = {5: (mid, u5, c5, d5), 4: (c5, u4, c4, d4), 3: (c4, u3, c3, d3), 2: (c3, u2, c2, d2), 1: (c2, u1, c1, d1)}
        for i, lvl in enumerate((5, 4, 3, 2, 1)):
            tin, u, c, skip = ups[lvl]
            wt = _np(sd["transposed_conv_%d.weight" % lvl]).astype(np.float32)          # [Cin][Cout][2][2]
            bt = _np(sd["transposed_conv_%d.bias" % lvl]).astype(np.float32)
            wt, bt = fold_bn(wt, bt, sd, "upsample_block_%d.0" % lvl, 1)
            src = f2.T_MID if i == 0 else f2.T_CU0 + i - 1
            n8 = tin // 8
            co = 8 if n8 % 8 == 0 else (4 if n8 % 4 == 0 else 2)
            chunks = f2.conv_chunks([(src, n8)], 1, 1, co)
            if u % 32 == 0 and not mt_over.get(L_UPT + i):
                # both dx of a 32-channel block in one workgroup: per dy a virtual output axis [block][dx][32 channels]
                w2 = []
                for dy in (0, 1):
                    wd = [np.ascontiguousarray(wt[:, :, dy, dx].T) for dx in (0, 1)]             # [u][tin]
                    w2.append(np.concatenate([wd[dx][b * 32:(b + 1) * 32] for b in range(u // 32) for dx in (0, 1)])[:, :, None, None])
                recipes[L_UPT + i] = (f2.build(w2, chunks, 1, 1, T(L_UPT + i), 4, f2.EPI_TC2), bt)
            else:
                w4 = [np.ascontiguousarray(wt[:, :, dy, dx].T)[:, :, None, None] for dy in (0, 1) for dx in (0, 1)]
                recipes[L_UPT + i] = (f2.build(w4, chunks, 1, 1, T(L_UPT + i), mt_over.get(L_UPT + i) or f2.pick_mt(u, tiles(lvl)), f2.EPI_TC), bt)
            w, b = conv_bn("conv_up_block_%d" % lvl)                                      # input = cat(up, skip_pre)
            recipes[L_UPC + i] = (f2.conv_layer(w, [(f2.T_UPT0 + i, u // 8), (f2.T_PRE0 + lvl - 1, skip // 8)], T(L_UPC + i), tiles(lvl - 1), mt=mt_over.get(L_UPC + i), gsize=gs_over.get(L_UPC + i), lds_target=LDS(L_UPC + i, lds_for(tiles(lvl - 1), c, 2 if i == 0 else 4)), **V(L_UPC + i)), b)
        # heads: the text + reconstruction row convolution is fused with its vertical sums (EPI_V; 56 + 77 -> 93 us: the 133 MB fp32 row buffer
        # is neither written nor read back); the output logit's is not (62 + 20 -> 87 us fused: its tiles of 10 finished rows cost more
        # row-convolution work than its 66 MB of rows; profiles/r04_heads_*.txt).  LM_FCN2_FUSED_HEADS: bit 0 = text / rec, bit 1 = output.
        fused = int(os.environ.get("LM_FCN2_FUSED_HEADS", "1"))
        head_epi, out_epi = (f2.EPI_V if fused & 1 else f2.EPI_T), (f2.EPI_V if fused & 2 else f2.EPI_T)
        wt, bt = conv_bn("conv_text_mask_out")
        wr, br = conv_bn("conv_reconstruct")
        rows = f2.text_rec_rows(wt, wr)
        recipes[L_TEXT] = (f2.build([rows], f2.conv_chunks([(f2.T_XUP, c1 // 8)], 1, 7, c1 // 8), 1, 7, T(L_TEXT), 1, head_epi, lds_target=LDS(L_TEXT), **V(L_TEXT)),
                           np.concatenate([np.zeros(16, np.float32), bt, br]))
        w, b = conv_bn("conv_pixels_1")
        px1_octets = int(os.environ.get("LM_FCN2_PX1_OCTETS", "2"))                     # experiments: feature octets per chunk of conv_pixels_1
        px_pdouble = [bool(int(v)) for v in os.environ.get("LM_FCN2_PX_PDOUBLE", "0,0").split(",")]      # ... double-buffered patch planes (px1, px2)
        recipes[L_PX1] = (f2.build([w], f2.pixel_chunks(f2.T_XUP, c1 // 8, f2.T_DP, 7, 7, octets=px1_octets), 7, 7, T(L_PX1), 2 if pm1 % 32 == 0 else 1, f2.EPI_PO, pdouble=px_pdouble[0], gsize=gs_over.get(L_PX1), lds_target=LDS(L_PX1), **V(L_PX1)), b)
        w, b = conv_bn("conv_pixels_2")
        recipes[L_PX2] = (f2.build([w], f2.pixel_chunks(f2.T_P1, pm1 // 8, f2.T_DP, 7, 7, octets=px_octets), 7, 7, T(L_PX2), 2 if pm2 % 32 == 0 else 1, f2.EPI_PO, pdouble=px_pdouble[1], gsize=gs_over.get(L_PX2), lds_target=LDS(L_PX2), **V(L_PX2)), b)
        w, b = conv_bn("conv_out")
        recipes[L_OUT] = (f2.build([f2.out_rows(w)], f2.pixel_chunks(f2.T_P2, pm2 // 8, f2.T_DP, 1, 7), 1, 7, T(L_OUT), 1, out_epi, pdouble=False, lds_target=LDS(L_OUT), **V(L_OUT)),
                          np.concatenate([np.zeros(16, np.float32), b]))
        # a tensor keeps its lo parts when a layer reading it runs the split format
        lo = np.zeros(f2.N_TENSORS, np.int32)
        for (desc, _, _, _), _ in recipes.values():
            if desc[2] in (2, 3):
                npl = int(desc[5] * desc[6])
                lo[desc[13:13 + 2 * npl:2]] = 1
        if self.handle2:
            self.lib.lm_fcn2_destroy(self.handle2)
        arr = (ctypes.c_int32 * 18)(*self.widths)
        self.handle2 = self.lib.lm_fcn2_create(arr, lo.ctypes.data, self.max_h, self.max_w)
        if not self.handle2:
            raise _lib.LecturemathError(_lib.LM_ERR_ARG, self.lib.last_error())
        self.recipes = {}
        for layer, ((desc, wpk, wblocks, need), bias) in recipes.items():
            bias = np.ascontiguousarray(bias, np.float32)
            self.lib.check(self.lib.lm_fcn2_set_layer(self.handle2, layer, desc.ctypes.data, desc.size, wpk.ctypes.data, wpk.nbytes, wblocks,
                                                      bias.ctypes.data, bias.size))
            self.recipes[layer] = {"kh": int(desc[0]), "kw": int(desc[1]), "terms": int(desc[2]), "mt": int(desc[3]), "chunks": int(desc[5]),
                                   "planes_per_chunk": int(desc[6]), "groups": int(desc[7]), "slices": int(desc[8]),
                                   "lds_bytes": int(need), "cout": int(desc[12]), "nc": int(desc[9]) & 15, "loader": (int(desc[9]) >> 8) & 1, "epilogue": int(desc[4]), "first_tensor": int(desc[13])}

    def executed_gflop(self, h, w):
        """MFMA flops the planar engine EXECUTES for one h x w frame (whole 16 x 16 tiles, whole 32-deep slices, three products per
        operand pair in the split-format layers), as opposed to the network's algorithmic flops"""
        from . import fcn2 as f2
        level_of = {f2.T_X0P: 0, f2.T_MID: 5, f2.T_XUP: 0, f2.T_DP: 0, f2.T_P1: 0, f2.T_P2: 0}
        for n in range(5):
            level_of[f2.T_PRE0 + n] = n
            level_of[f2.T_POOL0 + n] = n + 1
            level_of[f2.T_UPT0 + n] = 4 - n
        for n in range(4):
            level_of[f2.T_CU0 + n] = 4 - n
        total = 0.0
        for r in self.recipes.values():
            lv = level_of[r["first_tensor"]]
            tiles = (((h >> lv) + 15) // 16) * (((w >> lv) + 15) // 16)
            total += 2.0 * tiles * 256 * r["cout"] * r["slices"] * 32 * f2.FORMAT_PRODUCTS[r["terms"]] * (4 if r["epilogue"] == f2.EPI_TC else (2 if r["epilogue"] == f2.EPI_TC2 else 1))
        return total / 1e9

    def load_state_dict(self, sd):
        if self.planar:
            return self._load_planar(sd)
        d1, d2, d3, d4, d5, mid, u5, c5, u4, c4, u3, c3, u2, c2, u1, c1, pm1, pm2 = self.widths
        downs = [d1, d2, d3, d4, d5]

        def conv_bn(name):
            w, b = _np(sd[name + ".0.weight"]).astype(np.float32), _np(sd[name + ".0.bias"]).astype(np.float32)
            return fold_bn(w, b, sd, name + ".1", 0)

        def bias_pad(b):
            out = np.zeros(((len(b) + 31) // 32) * 32, np.float32)
            out[:len(b)] = b
            return out

        def ck_for(*chans):
            return 16 if all(c % 16 == 0 for c in chans) else 8

        h = self.precision != "fp32"
        hck = {"f16x3": 0, "f16x2": -2, "f16": -1}.get(self.precision, 0)       # lm_fcn.hip: products per operand pair

        def mfma(w, cin_map, cin_padded):
            """(packed weights, ck): ck = 0 selects the fp16-split kernel"""
            if h:
                return pack_mfma_h(w, cin_map, sum(cin_padded) if isinstance(cin_padded, tuple) else cin_padded), hck
            ck = ck_for(cin_padded) if not isinstance(cin_padded, tuple) else ck_for(*cin_padded)
            return pack_mfma(w, ck, cin_map, cin_padded if not isinstance(cin_padded, tuple) else sum(cin_padded)), ck

        # encoder + mid (layer 1 sees the 3 RGB channels padded to 8)
        cin = [3] + downs
        for n in range(5):
            w, b = conv_bn("conv_down_block_%d" % (n + 1))
            cpad = 8 if n == 0 else cin[n]
            wpk, ck = mfma(w, range(cin[n]), cpad)
            self._set(L_DOWN + n, wpk, bias_pad(b), cpad, downs[n], self.kk, ck)
        w, b = conv_bn("mid_block")
        wpk, ck = mfma(w, range(d5), d5)
        self._set(L_MID, wpk, bias_pad(b), d5, mid, self.kk, ck)
        # decoder: level 5 .. 1
        ups = {5: (mid, u5, c5, d5), 4: (c5, u4, c4, d4), 3: (c4, u3, c3, d3), 2: (c3, u2, c2, d2), 1: (c2, u1, c1, d1)}
        for i, lvl in enumerate((5, 4, 3, 2, 1)):
            tin, u, c, skip = ups[lvl]
            wt = _np(sd["transposed_conv_%d.weight" % lvl]).astype(np.float32)          # [Cin][Cout][2][2]
            bt = _np(sd["transposed_conv_%d.bias" % lvl]).astype(np.float32)
            wt, bt = fold_bn(wt, bt, sd, "upsample_block_%d.0" % lvl, 1)
            if h:       # one launch: the four (dy, dx) sets are the four "taps" of the packing (lm_k_convT_mfma_h)
                self._set(L_UPT + i, pack_mfma_h(np.ascontiguousarray(wt.transpose(1, 0, 2, 3)), range(tin), tin), bias_pad(bt), tin, u, 1, hck)
            else:
                sets = [mfma(np.ascontiguousarray(wt[:, :, dy, dx].T)[:, :, None, None], range(tin), tin) for dy in (0, 1) for dx in (0, 1)]
                self._set(L_UPT + i, np.concatenate([p for p, _ in sets]), bias_pad(bt), tin, u, 1, sets[0][1])
            w, b = conv_bn("conv_up_block_%d" % lvl)                                      # input = cat(up, skip_pre)
            wpk, ck = mfma(w, range(u + skip), (u, skip))
            self._set(L_UPC + i, wpk, bias_pad(b), u + skip, c, self.kk, ck)
        # heads: inputs are (diff | features | zero pad) buffers
        s0, s1, s2 = _pad8(3 + c1), _pad8(3 + pm1), _pad8(3 + pm2)
        # The fp16-split formats with the shipped kernel sizes (7x7 pixel branch, 3x3 elsewhere) run the heads on the MFMA path
        # (row convolution + vertical sum) and keep x_up1 / diff / pixel features in buffers of their own: a (diff, features)
        # input is the two-input concatenation [d0 d1 d2 0 | features].  Everything else: the round-1 layout, one
        # (diff | features | pad) buffer per stage and VALU kernels for the heads.
        if h and self.pk == 7 and self.kk == 3:
            def head_bias(b):       # [0..31]: zeros for the row convolution's epilogue, [32..]: the bias lm_k_vsum adds
                out = np.zeros(64, np.float32)
                out[32:32 + len(b)] = b
                return out

            def cat_map(nfeat):     # weight input channel -> logical channel of [diff(3) 0 | features]
                return [0, 1, 2] + list(range(4, 4 + nfeat))

            wt, bt = conv_bn("conv_text_mask_out")
            wr, br = conv_bn("conv_reconstruct")
            self._set(L_TEXT, pack_text_rec_rows_h(wt, wr, range(c1), c1), head_bias(np.concatenate([bt, br])), c1, 4, self.pk, hck)
            w, b = conv_bn("conv_pixels_1")
            self._set(L_PX1, pack_mfma_h(w, cat_map(c1), 4 + c1), bias_pad(b), 4 + c1, pm1, self.pk, hck)
            w, b = conv_bn("conv_pixels_2")
            self._set(L_PX2, pack_mfma_h(w, cat_map(pm1), 4 + pm1), bias_pad(b), 4 + pm1, pm2, self.pk, hck)
            w, b = conv_bn("conv_out")
            self._set(L_OUT, pack_rows_h(w, cat_map(pm2), 4 + pm2), head_bias(b), 4 + pm2, 1, self.pk, hck)
            return
        w, b = conv_bn("conv_text_mask_out")
        self._set(L_TEXT, pack_small(w, range(3, 3 + c1), s0), np.pad(b, (0, 4 - len(b))), s0, 1, self.pk, 8)
        w, b = conv_bn("conv_reconstruct")
        self._set(L_REC, pack_small(w, range(3, 3 + c1), s0), np.pad(b, (0, 4 - len(b))), s0, 3, self.kk, 8)
        w, b = conv_bn("conv_pixels_1")
        wpk, ck = (pack_mfma_h(w, range(3 + c1), s0), hck) if h else (pack_mfma(w, 8, range(3 + c1), s0), 8)
        self._set(L_PX1, wpk, bias_pad(b), s0, pm1, self.pk, ck)
        w, b = conv_bn("conv_pixels_2")
        wpk, ck = (pack_mfma_h(w, range(3 + pm1), s1), hck) if h else (pack_mfma(w, 8, range(3 + pm1), s1), 8)
        self._set(L_PX2, wpk, bias_pad(b), s1, pm2, self.pk, ck)
        w, b = conv_bn("conv_out")
        self._set(L_OUT, pack_small(w, range(3 + pm2), s2), np.pad(b, (0, 4 - len(b))), s2, 1, self.pk, 8)

    def set_layer_precision(self, layer, precision):
        """Operand format of ONE layer ("f16x3" / "f16x2" / "f16"); the engine must have been loaded with an fp16-split precision."""
        terms = {"f16x3": 3, "f16x2": 2, "f16": 1}[precision]
        if self.planar:
            raise _lib.LecturemathError(_lib.LM_ERR_STATE, "the planar engine's formats are fixed by load_state_dict (precision=...)")
        self.lib.check(self.lib.lm_fcn_set_layer_terms(self.handle, int(layer), terms))

    def forward_raw(self, rgb_ptr, h, w, out_ptr, text_ptr=None, rec_ptr=None, stream=None):
        """one forward pass on raw device addresses (uint8 [h,w,3] in; fp32 logit [h,w], text logit [h,w], rec [3,h,w] out, each optional)
        on `stream` (a HIP stream handle; default: the backend's current stream)"""
        fwd, hd = (self.lib.lm_fcn2_forward, self.handle2) if self.planar else (self.lib.lm_fcn_forward, self.handle)
        if not hd:
            raise _lib.LecturemathError(_lib.LM_ERR_STATE, "FcnEngine.forward before load_state_dict")
        self.lib.check(fwd(hd, rgb_ptr, h, w, out_ptr, text_ptr, rec_ptr, self.be.stream() if stream is None else stream))

    def forward(self, rgb):
        """rgb: device (or host numpy) uint8 [H,W,3] -> device fp32 (logit [H,W], text logit [H,W], rec [3,H,W])."""
        if isinstance(rgb, np.ndarray):
            rgb = self.be.from_host(rgb)
        h, w = int(rgb.shape[0]), int(rgb.shape[1])
        out = self.be.empty((h, w), np.float32)
        text = self.be.empty((h, w), np.float32)
        rec = self.be.empty((3, h, w), np.float32)
        self.forward_raw(_lib.ptr(rgb), h, w, _lib.ptr(out), _lib.ptr(text), _lib.ptr(rec))
        return out, text, rec
