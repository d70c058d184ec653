"""FCN_LectureNet inference on the MI355X with the reference's API (lecturenet_v1/FCN_lecturenet.py):
CreateFromConfig :620-659, load_state_dict / eval / cuda (torch.nn.Module methods the callers use,
pre_ST3D_v3.0_01_binarize.py:30-37, test_FCN_binarizer.py:38-46), binarize :430-505, prepare_image :607-618,
from_img_space_to_cv2 :534-555.  The convolution stack runs in liblecturemath_hip.so (lm_fcn.hip); training-only members
of the reference class are not provided (inference path only, SURVEY.md section 2 #7b/#23)."""
import numpy as np
import PIL.Image

from lecturemath_amd import _lib, fcn


class FCN_LectureNet:
    MAX_PIXELS = 2500000      # :435

    def __init__(self, channels, n_conv_down_1, n_conv_down_2, n_conv_down_3, n_conv_down_4, n_conv_down_5, mid_block,
                 n_upsample_5, n_conv_up_5, n_upsample_4, n_conv_up_4, n_upsample_3, n_conv_up_3, n_upsample_2, n_conv_up_2,
                 n_upsample_1, n_conv_up_1, kernel_size, n_pmaps_1, n_pmaps_2, pixel_kernel_size, reconstruction_mode):
        if channels != 3 or reconstruction_mode:
            raise NotImplementedError("the MI355X path implements the 3-channel binarization branch (reconstruction_mode=False)")
        self.widths = [n_conv_down_1, n_conv_down_2, n_conv_down_3, n_conv_down_4, n_conv_down_5, mid_block, n_upsample_5,
                       n_conv_up_5, n_upsample_4, n_conv_up_4, n_upsample_3, n_conv_up_3, n_upsample_2, n_conv_up_2, n_upsample_1,
                       n_conv_up_1, n_pmaps_1, n_pmaps_2]
        self.kernel_size, self.pixel_kernel_size = kernel_size, pixel_kernel_size
        self.reconstruction_mode = reconstruction_mode
        self._sd = None
        self._engine = None
        self._engine_hw = (0, 0)

    # ---- torch.nn.Module look-alikes used by the callers
    def load_state_dict(self, state_dict, strict=True):
        self._sd = dict(state_dict)
        if self._engine is not None:
            self._engine.load_state_dict(self._sd)
        if getattr(self, "_engine2", None) is not None:         # the second engine of binarize_frames_device is rebuilt on next use
            self._engine2.close()
            self._engine2 = None

    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def cpu(self):
        raise _lib.LecturemathLibraryError("FCN_LectureNet of lecturemath_amd runs on the GPU only (no CPU fallback)")

    def parameters(self):
        return iter(())

    def _get_engine(self, h, w):
        if self._sd is None:
            raise RuntimeError("load_state_dict() must be called before binarize()")
        if self._engine is None or h > self._engine_hw[0] or w > self._engine_hw[1] or h * w > self._engine_hw[0] * self._engine_hw[1]:
            if self._engine is not None:
                self._engine.close()
            self._engine = fcn.FcnEngine(self.widths, self.pixel_kernel_size, self.kernel_size, h, w)
            self._engine.load_state_dict(self._sd)
            self._engine_hw = (h, w)
        return self._engine

    # ---- inference
    def forward_logits(self, rgb_u8):
        """Extension: uint8 RGB [H,W,3] (numpy or device tensor) -> device fp32 (logit, text logit, reconstruction)."""
        return self._get_engine(int(rgb_u8.shape[0]), int(rgb_u8.shape[1])).forward(rgb_u8)

    def _resizer(self):
        if getattr(self, "_rs", None) is None:
            from lecturemath_amd import resize
            self._rs = resize.DeviceResizer()
        return self._rs

    def _halve_on_device(self, rgb_dev, width, height):
        """:434-437 -- `while width * height > MAX_PIXELS: resize((int(width / 2), int(height / 2)), LANCZOS)` on the device
        (Pillow's two-pass fixed-point resampling, lecturemath_amd/resize.py)"""
        while width * height > FCN_LectureNet.MAX_PIXELS:
            width, height = int(width / 2), int(height / 2)
            rgb_dev = self._resizer().lanczos(rgb_dev, width, height)
        return rgb_dev, width, height

    def binarize_frames_device(self, rgb_frames, binary_threshold=128):
        """Extension for whole videos: uint8 RGB [n,H,W,3] (numpy or device tensor) -> device uint8 [n,H,W], the worker's inverted
        binary (ink = 255; :452-467 + FCN_lecturenet_binarizer.py:54); nothing leaves the device.  Frames above 2.5 MP go through the
        reference's resize branch (:434-437 LANCZOS halving, :481-486 INTER_NEAREST back) on the device as well."""
        n, h, w = int(rgb_frames.shape[0]), int(rgb_frames.shape[1]), int(rgb_frames.shape[2])
        big = w * h > FCN_LectureNet.MAX_PIXELS
        nw, nh = w, h
        while nw * nh > FCN_LectureNet.MAX_PIXELS:
            nw, nh = int(nw / 2), int(nh / 2)
        eng = self._get_engine(nh, nw)
        lib, be = eng.lib, eng.be
        out = be.empty((n, h, w), np.uint8)
        # Two engines on two HIP streams, frames dealt alternately: the layers below 1/8 resolution launch 288-480 workgroups on 256 CUs and a
        # second forward pass in flight fills what one leaves idle (profiles/r04_fcn_two_streams.txt: 437 -> 475 frames/s).  Each engine has
        # its own activation arena; everything a frame touches is enqueued on its engine's stream.
        engines, streams = [eng], [None]
        if be.device and n >= 2 and eng.planar:
            torch = be.torch
            if getattr(self, "_engine2", None) is None or self._engine2_hw != self._engine_hw:
                if getattr(self, "_engine2", None) is not None:
                    self._engine2.close()
                self._engine2 = fcn.FcnEngine(self.widths, self.pixel_kernel_size, self.kernel_size, self._engine_hw[0], self._engine_hw[1])
                self._engine2.load_state_dict(self._sd)
                self._engine2_hw = self._engine_hw
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream()
            self._side_stream.wait_stream(torch.cuda.current_stream())
            engines.append(self._engine2)
            streams.append(self._side_stream)

        def one(i, e):
            small = be.empty((nh, nw), np.uint8) if big else None
            frame = rgb_frames[i] if not isinstance(rgb_frames, np.ndarray) else be.from_host(rgb_frames[i])
            if big:
                frame, _, _ = self._halve_on_device(frame, w, h)
            logits, _, _ = e.forward(frame)
            dst = out[i] if be.device else out[i:i + 1]
            if big:
                lib.check(lib.lm_threshold_invert(_lib.ptr(logits), _lib.ptr(small), nh * nw, int(binary_threshold), be.stream()))
                lib.check(lib.lm_upsample_nearest_u8(_lib.ptr(small), nh, nw, 1, _lib.ptr(dst), h, w, be.stream()))
            else:
                lib.check(lib.lm_threshold_invert(_lib.ptr(logits), _lib.ptr(dst), h * w, int(binary_threshold), be.stream()))
        for i in range(n):
            k = i % len(engines)
            if streams[k] is None:
                one(i, engines[k])
            else:
                with be.torch.cuda.stream(streams[k]):
                    one(i, engines[k])
        if len(engines) > 1:
            be.torch.cuda.current_stream().wait_stream(self._side_stream)
        return out

    def binarize(self, PIL_image, return_others=False, force_binary=False, binary_treshold=128, apply_sigmoid=True):
        o_width, o_height = PIL_image.size
        width, height = o_width, o_height
        if width * height > FCN_LectureNet.MAX_PIXELS and PIL_image.mode not in ("RGB", "L"):
            # the reference resizes the image in ITS OWN mode and converts afterwards (:434-437); Pillow resamples palette / bilevel images
            # with NEAREST and premultiplies alpha, which the device resizer (RGB / L bytes) does not restate: those modes take Pillow's path
            from PIL import Image
            while width * height > FCN_LectureNet.MAX_PIXELS:
                PIL_image = PIL_image.resize((int(width / 2), int(height / 2)), Image.LANCZOS)
                width, height = PIL_image.size
        rgb_full = np.asarray(PIL_image.convert("RGB"), dtype=np.uint8)
        eng0_be = None
        if width * height > FCN_LectureNet.MAX_PIXELS:
            # the whole > 2.5 MP branch on the device: one upload of the frame, LANCZOS halving(s), network, threshold, NEAREST enlargement
            from lecturemath_amd.device import Backend
            eng0_be = Backend(_lib.load())
            rgb, width, height = self._halve_on_device(eng0_be.from_host(rgb_full), width, height)
        else:
            rgb = rgb_full
        eng = self._get_engine(height, width)
        out, text, rec = eng.forward(rgb)
        lib, be = eng.lib, eng.be
        n = height * width
        resized = o_width != width
        if resized and not force_binary:
            raise NotImplementedError("INTER_CUBIC upsampling of non-binary outputs (:487-492) is not implemented")

        def enlarge(dev_u8):
            return self._resizer().nearest(dev_u8, o_width, o_height) if resized else dev_u8

        def post(logits):
            if force_binary and apply_sigmoid:
                dst = be.empty((height, width), np.uint8)
                lib.check(lib.lm_threshold(_lib.ptr(logits), _lib.ptr(dst), n, int(binary_treshold), 0, be.stream()))
                return be.to_host(enlarge(dst))
            v = be.to_host(logits)
            if apply_sigmoid:
                v = (1.0 / (1.0 + np.exp(-v, dtype=np.float32))).astype(np.float32)
            img = (v * 255).astype(np.uint8)
            if force_binary:
                img[img >= binary_treshold] = 255
                img[img < binary_treshold] = 0
            return be.to_host(enlarge(be.from_host(img))) if resized else img

        binary = post(out)
        text_mask = rec_img = None
        if return_others:
            text_mask = post(text)
            rec_img = self.from_img_space_to_cv2(be.to_host(rec))
            if resized:
                rec_img = be.to_host(self._resizer().nearest(be.from_host(rec_img), o_width, o_height))
        return (binary, text_mask, rec_img) if return_others else binary

    def from_img_space_to_cv2(self, image):
        img = np.transpose(np.array(image, dtype=np.float32, copy=True), (1, 2, 0))
        img *= 0.5
        img += 0.5
        img = np.ascontiguousarray(img[:, :, ::-1])
        img *= 255
        img[img > 255] = 255
        img[img < 0] = 0
        return img.astype(np.uint8)

    @staticmethod
    def prepare_image(PIL_image):
        import torch
        a = np.asarray(PIL_image.convert("RGB"), dtype=np.uint8)
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)
        return ((t - 0.5) / 0.5).unsqueeze(0)

    @staticmethod
    def CreateFromConfig(config, in_channels, reconstruction_mode):
        w = [config.get(key, default) for key, default in fcn.WIDTH_KEYS]
        pix_k = config.get("FCN_BINARIZER_NET_PIXEL_KERNEL_SIZE", 3)
        k = config.get("FCN_BINARIZER_NET_KERNEL_SIZE", 3)
        return FCN_LectureNet(in_channels, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9], w[10], w[11], w[12], w[13],
                              w[14], w[15], k, w[16], w[17], pix_k, reconstruction_mode)
