"""FCN-LectureNet forward + binarize post-processing, CPU fp32 -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

A torch.nn.functional restatement (no nn.Module) of AccessMath/lecturenet_v1/FCN_lecturenet.py working directly
from a state_dict with the reference's keys:
    encode_decode :260-323, forward (non-reconstruction branch) :364-403, binarize :430-505,
    prepare_image :607-618, from_img_space_to_cv2 :534-555.
Floating-point: this is the "plain PyTorch fp32 reference" for the HIP conv stack (tolerance 1e-3 on logits).
Parity status: pinned against the reference module itself (tests/golden/make_golden_fcn.py -> g5_fcn_*.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _bn(sd, p, x):
    return F.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"), _t(sd, p + ".bias"),
                        training=False, eps=BN_EPS)


def _block(sd, name, x, act="gelu"):
    w = _t(sd, name + ".0.weight")
    x = F.conv2d(x, w, _t(sd, name + ".0.bias"), stride=1, padding=(w.shape[2] - 1) // 2)
    x = _bn(sd, name + ".1", x)
    if act == "gelu":
        return F.gelu(x)
    if act == "tanh":
        return torch.tanh(x)
    return x


def _up(sd, n, x, out_hw):
    w = _t(sd, "transposed_conv_%d.weight" % n)
    y = F.conv_transpose2d(x, w, _t(sd, "transposed_conv_%d.bias" % n), stride=2, padding=0,
                           output_padding=(out_hw[0] - 2 * x.shape[2], out_hw[1] - 2 * x.shape[3]))
    return F.gelu(_bn(sd, "upsample_block_%d.0" % n, y))


def forward(sd, x0, return_intermediates=False):
    """x0: [1,3,H,W] fp32 in [-1,1].  Returns (binarization logit, text-mask logit, reconstruction) like forward()."""
    inter = {}
    pre, x = [], x0
    for n in range(1, 6):
        p = _block(sd, "conv_down_block_%d" % n, x)
        pre.append(p)
        x = F.max_pool2d(p, 2)
        inter["down%d_pre" % n] = p
    x = _block(sd, "mid_block", x)
    inter["mid"] = x
    for n in range(5, 0, -1):
        skip = pre[n - 1]
        u = _up(sd, n, x, skip.shape[2:])
        x = _block(sd, "conv_up_block_%d" % n, torch.cat((u, skip), 1))
        inter["up%d" % n] = x
    x_up1 = x
    text = _block(sd, "conv_text_mask_out", x_up1, act=None)
    rec = _block(sd, "conv_reconstruct", x_up1, act="tanh")
    diff = (x0 - rec) * torch.sigmoid(text)
    p1 = _block(sd, "conv_pixels_1", torch.cat((diff, x_up1), 1))
    p2 = _block(sd, "conv_pixels_2", torch.cat((diff, p1), 1))
    out = _block(sd, "conv_out", torch.cat((diff, p2), 1), act=None)
    inter.update(text=text, rec=rec, diff=diff, p1=p1, p2=p2)
    return (out, text, rec, inter) if return_intermediates else (out, text, rec)


def prepare_image(rgb_u8):
    """prepare_image :607-618 on an HxWx3 uint8 RGB array."""
    t = torch.from_numpy(np.ascontiguousarray(rgb_u8.transpose(2, 0, 1))).to(torch.float32).div(255)
    return ((t - 0.5) / 0.5).unsqueeze(0)


def rec_to_bgr_u8(rec_chw):
    """from_img_space_to_cv2 :534-555."""
    img = np.transpose(np.array(rec_chw, dtype=np.float32, copy=True), (1, 2, 0))
    img *= 0.5
    img += 0.5
    img = img[:, :, ::-1].copy()
    img *= 255
    img[img > 255] = 255
    img[img < 0] = 0
    return img.astype(np.uint8)


def binarize(sd, rgb_u8, thr=128):
    """binarize(return_others=True, force_binary=True) :430-505 for images <= 2.5 MP (no resize branch):
    returns (binary {0,255}, text_mask {0,255}, rec BGR uint8) -- NOT inverted (the worker inverts)."""
    with torch.no_grad():
        out, text, rec = forward(sd, prepare_image(rgb_u8))
        res = torch.sigmoid(out)
        tm = torch.sigmoid(text)
    b = (res[0, 0].numpy() * 255).astype(np.uint8)
    b[b >= thr] = 255
    b[b < thr] = 0
    t = (tm[0, 0].numpy() * 255).astype(np.uint8)
    t[t >= thr] = 255
    t[t < thr] = 0
    return b, t, rec_to_bgr_u8(rec[0].numpy())


def random_state_dict(widths, pixel_kernel=7, kernel=3, seed=0):
    """State dict with the reference's keys/shapes, xavier-normal conv weights and
    randomised BN statistics/affine so that the BN fold is exercised.
    widths = (d1,d2,d3,d4,d5, mid, u5,c5,u4,c4,u3,c3,u2,c2,u1,c1, pm1, pm2)."""
    g = torch.Generator().manual_seed(seed)
    d1, d2, d3, d4, d5, mid, u5, c5, u4, c4, u3, c3, u2, c2, u1, c1, pm1, pm2 = widths
    sd = {}

    def conv(name, cin, cout, k):
        fan_in, fan_out = cin * k * k, cout * k * k
        std = (2.0 / (fan_in + fan_out)) ** 0.5
        sd[name + ".weight"] = torch.randn((cout, cin, k, k), generator=g) * std
        sd[name + ".bias"] = (torch.rand(cout, generator=g) - 0.5) * 0.2

    def bn(name, c):
        sd[name + ".weight"] = 0.5 + torch.rand(c, generator=g)
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)

    def block(name, cin, cout, k):
        conv(name + ".0", cin, cout, k)
        bn(name + ".1", cout)

    downs = [3, d1, d2, d3, d4, d5]
    for n in range(1, 6):
        block("conv_down_block_%d" % n, downs[n - 1], downs[n], kernel)
    block("mid_block", d5, mid, kernel)
    ups = {5: (mid, u5, c5, d5), 4: (c5, u4, c4, d4), 3: (c4, u3, c3, d3), 2: (c3, u2, c2, d2), 1: (c2, u1, c1, d1)}
    for n in range(5, 0, -1):
        cin, u, c, skip = ups[n]
        std = (2.0 / (cin * 4 + u * 4)) ** 0.5
        sd["transposed_conv_%d.weight" % n] = torch.randn((cin, u, 2, 2), generator=g) * std
        sd["transposed_conv_%d.bias" % n] = (torch.rand(u, generator=g) - 0.5) * 0.2
        bn("upsample_block_%d.0" % n, u)
        block("conv_up_block_%d" % n, u + skip, c, kernel)
    block("conv_pixels_1", 3 + c1, pm1, pixel_kernel)
    block("conv_pixels_2", 3 + pm1, pm2, pixel_kernel)
    block("conv_out", 3 + pm2, 1, pixel_kernel)
    block("conv_text_mask_out", c1, 1, pixel_kernel)
    block("conv_reconstruct", c1, 3, kernel)
    return sd


SHIPPED_WIDTHS = (48, 96, 192, 384, 768, 768, 384, 384, 192, 192, 96, 96, 48, 48, 32, 32, 32, 16)   # configs/FCN_LectureNet.conf:109-132
