"""Deterministic synthetic whiteboard inputs (no video / dataset is available offline).

Generators follow the recipes fixed in SURVEY.md section 8(d):
  * whiteboard_rgb()      config 1/2/5: light board + illumination ramp + dark glyph strokes (RGB uint8)
  * binary_stream()       config 3/4: persistent canvas, glyphs added over time, partial erasures,
                          one transient blob per frame, occasional 1-px jitter, optional occluder
  * logits_from_binary()  fp32 "FCN logits" (+-4 around the mask) so that the threshold kernel is
                          exercised: ink (255 in the step-01 output) <=> NEGATIVE logit, because the
                          worker inverts the thresholded sigmoid (FCN_lecturenet_binarizer.py:54).
Pure numpy; shared by tests and bench.py.
"""
import numpy as np


def _draw_glyph(canvas, x, y, gw, gh, thick, kind, value=255):
    """kind 0: box outline, 1: L (left + bottom), 2: inverted L (top + right), 3: plus/cross."""
    h, w = canvas.shape
    x1, y1 = min(x + gw, w), min(y + gh, h)
    if kind == 0:
        canvas[y:min(y + thick, y1), x:x1] = value
        canvas[max(y1 - thick, y):y1, x:x1] = value
        canvas[y:y1, x:min(x + thick, x1)] = value
        canvas[y:y1, max(x1 - thick, x):x1] = value
    elif kind == 1:
        canvas[y:y1, x:min(x + thick, x1)] = value
        canvas[max(y1 - thick, y):y1, x:x1] = value
    elif kind == 2:
        canvas[y:min(y + thick, y1), x:x1] = value
        canvas[y:y1, max(x1 - thick, x):x1] = value
    else:
        cx, cy = x + gw // 2, y + gh // 2
        canvas[y:y1, cx:min(cx + thick, x1)] = value
        canvas[cy:min(cy + thick, y1), x:x1] = value


def _random_glyphs(rng, n, h, w, margin=10, min_ext=6, max_ext=28):
    gw = rng.integers(min_ext, max_ext + 1, n)
    gh = rng.integers(min_ext, max_ext + 1, n)
    x = rng.integers(margin, max(margin + 1, w - margin - max_ext), n)
    y = rng.integers(margin, max(margin + 1, h - margin - max_ext), n)
    thick = rng.integers(2, 4, n)
    kind = rng.integers(0, 4, n)
    return np.stack([x, y, gw, gh, thick, kind], axis=1)


def glyph_mask(h=1080, w=1920, n_glyphs=1500, seed=20211):
    """uint8 {0,255} ink mask of n_glyphs strokes (config 1's synthetic *binary* frame)."""
    rng = np.random.default_rng(seed)
    m = np.zeros((h, w), np.uint8)
    for g in _random_glyphs(rng, n_glyphs, h, w):
        _draw_glyph(m, *[int(v) for v in g])
    return m


def whiteboard_rgb(h=1080, w=1920, n_glyphs=1500, seed=20211):
    """RGB uint8 whiteboard-like frame + its ink mask."""
    rng = np.random.default_rng(seed)
    ramp = np.linspace(-10.0, 10.0, w, dtype=np.float32)[None, :, None]
    img = 235.0 + rng.normal(0.0, 3.0, (h, w, 3)).astype(np.float32) + ramp
    mask = np.zeros((h, w), np.uint8)
    ink = np.zeros((h, w), np.uint8)
    for g in _random_glyphs(rng, n_glyphs, h, w):
        gi = [int(v) for v in g]
        _draw_glyph(mask, *gi)
        _draw_glyph(ink, *gi, value=int(rng.integers(20, 81)))
    img = np.clip(img, 0, 255)
    img[mask > 0] = ink[mask > 0][:, None]
    return img.astype(np.uint8), mask


def whiteboard_stream(n_frames, h=1080, w=1920, glyphs_start=400, glyphs_per_frame=12, seed=20212):
    """Yield n_frames RGB uint8 frames of ONE evolving whiteboard (fixed background noise / illumination ramp, ink strokes
    accumulating over time) for the RGB -> FCN -> ... end-to-end measurements: consecutive frames share most of their
    content, like sampled lecture video."""
    rng = np.random.default_rng(seed)
    ramp = np.linspace(-10.0, 10.0, w, dtype=np.float32)[None, :, None]
    board = np.clip(235.0 + rng.normal(0.0, 3.0, (h, w, 3)).astype(np.float32) + ramp, 0, 255).astype(np.uint8)
    total = glyphs_start + glyphs_per_frame * n_frames
    glyphs = _random_glyphs(rng, total, h, w)
    ink = rng.integers(20, 81, total)
    mask = np.zeros((h, w), np.uint8)
    done = 0
    for t in range(n_frames):
        upto = glyphs_start + glyphs_per_frame * (t + 1)
        for g, v in zip(glyphs[done:upto], ink[done:upto]):
            _draw_glyph(mask, *[int(x) for x in g], value=int(v))
        done = upto
        frame = board.copy()
        frame[mask > 0] = mask[mask > 0][:, None]
        yield frame


def binary_stream(n_frames, h=1080, w=1920, seed=20213, glyphs_per_add=40, add_every=2, erase_every=250,
                  jitter_p=0.02, jitter_frac=0.05, transient=True, occluder=False, max_ext=28):
    """Yield n_frames uint8 {0,255} frames (255 = ink), a persistent board evolving in time."""
    rng = np.random.default_rng(seed)
    canvas = np.zeros((h, w), np.uint8)
    glyphs = np.zeros((0, 6), np.int64)
    next_erase = erase_every + int(rng.integers(-erase_every // 10, erase_every // 10 + 1)) if erase_every else -1
    occ_w, occ_h, occ_speed = max(w // 6, 8), max(h // 2, 8), max(w // 240, 1)
    for t in range(n_frames):
        if add_every and t % add_every == 0:
            new = _random_glyphs(rng, glyphs_per_add, h, w, max_ext=max_ext)
            for g in new:
                _draw_glyph(canvas, *[int(v) for v in g])
            glyphs = np.concatenate([glyphs, new], axis=0)
        if erase_every and t == next_erase:
            frac = (2, 3)[int(rng.integers(0, 2))]
            span = w // frac
            x0 = int(rng.integers(0, w - span + 1))
            canvas[:, x0:x0 + span] = 0
            keep = (glyphs[:, 0] + glyphs[:, 2] <= x0) | (glyphs[:, 0] >= x0 + span)
            glyphs = glyphs[keep]
            next_erase = t + erase_every + int(rng.integers(-erase_every // 10, erase_every // 10 + 1))
        frame = canvas.copy()
        if jitter_p and len(glyphs) and rng.random() < jitter_p:
            sel = rng.choice(len(glyphs), max(1, int(len(glyphs) * jitter_frac)), replace=False)
            for g in glyphs[sel]:
                x, y, gw, gh, th, kd = [int(v) for v in g]
                frame[y:y + gh, x:x + gw] = 0
                dx, dy = int(rng.integers(-1, 2)), int(rng.integers(-1, 2))
                _draw_glyph(frame, max(x + dx, 0), max(y + dy, 0), gw, gh, th, kd)
        if transient:
            bx, by = int(rng.integers(0, max(w - 6, 1))), int(rng.integers(0, max(h - 6, 1)))
            frame[by:by + 6, bx:bx + 6] = 255
        if occluder:
            ox = (t * occ_speed) % (w + occ_w) - occ_w
            oy = (h - occ_h) // 2
            frame[oy:oy + occ_h, max(ox, 0):max(ox + occ_w, 0)] = 0
        yield frame


def logits_from_binary(binary, seed=0, margin=4.0, noise=0.5):
    """fp32 logits whose sigmoid->threshold->invert reproduces `binary` (255 = ink <=> negative logit)."""
    rng = np.random.default_rng(seed)
    lg = np.where(binary > 0, -margin, margin).astype(np.float32)
    lg += (rng.random(binary.shape, dtype=np.float32) - 0.5) * (2.0 * noise)
    return lg


# FCN-LectureNet at the shipped configuration (configs/FCN_LectureNet.conf:109-132):
# (down 1..5, mid, then (transposed conv, conv) widths of up blocks 5..1, pixel branch 1, 2)
FCN_SHIPPED_WIDTHS = (48, 96, 192, 384, 768, 768, 384, 384, 192, 192, 96, 96, 48, 48, 32, 32, 32, 16)


def fcn_random_state_dict(widths=FCN_SHIPPED_WIDTHS, pixel_kernel=7, kernel=3, seed=0):
    """Random-init weights with the reference network's state_dict keys and shapes (there are no checkpoints offline): scaled
    normal conv weights, small biases, BatchNorm statistics away from (0, 1) so that folding them matters."""
    import torch
    gen = torch.Generator().manual_seed(seed)
    d1, d2, d3, d4, d5, mid, u5, c5, u4, c4, u3, c3, u2, c2, u1, c1, pm1, pm2 = widths
    sd = {}

    def add_conv(name, shape, fan):
        sd[name + ".weight"] = torch.randn(shape, generator=gen) * (2.0 / fan) ** 0.5
        sd[name + ".bias"] = (torch.rand(shape[0] if name.startswith("conv") or name.startswith("mid") else shape[1], generator=gen) - 0.5) * 0.2

    def add_bn(name, c):
        sd[name + ".weight"] = 0.5 + torch.rand(c, generator=gen)
        sd[name + ".bias"] = torch.randn(c, generator=gen) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=gen) * 0.1
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=gen)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)

    def add_block(name, cin, cout, k):
        add_conv(name + ".0", (cout, cin, k, k), (cin + cout) * k * k)
        add_bn(name + ".1", cout)

    chain = [3, d1, d2, d3, d4, d5]
    for n in range(1, 6):
        add_block("conv_down_block_%d" % n, chain[n - 1], chain[n], kernel)
    add_block("mid_block", d5, mid, kernel)
    for n, cin, up, out, skip in ((5, mid, u5, c5, d5), (4, c5, u4, c4, d4), (3, c4, u3, c3, d3), (2, c3, u2, c2, d2), (1, c2, u1, c1, d1)):
        add_conv("transposed_conv_%d" % n, (cin, up, 2, 2), (cin + up) * 4)
        add_bn("upsample_block_%d.0" % n, up)
        add_block("conv_up_block_%d" % n, up + skip, out, kernel)
    add_block("conv_pixels_1", 3 + c1, pm1, pixel_kernel)
    add_block("conv_pixels_2", 3 + pm1, pm2, pixel_kernel)
    add_block("conv_out", 3 + pm2, 1, pixel_kernel)
    add_block("conv_text_mask_out", c1, 1, pixel_kernel)
    add_block("conv_reconstruct", c1, 3, kernel)
    return sd

