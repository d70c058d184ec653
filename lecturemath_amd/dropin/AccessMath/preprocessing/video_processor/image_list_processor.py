"""Frame source for exported lectures: a folder of images driving the video-worker protocol
(AccessMath/preprocessing/video_processor/image_list_processor.py: ImageListGenerator :7-80, ImageListProcessor :82-199;
wired by ConsoleUIProcess.start_image_list_preprocessing, console_ui_process.py:188-221).  Same class names, constructor
arguments, `force_resolution` and `doProcessing(video_worker, limit, verbose)`; the worker sees exactly the calls the reference
makes: initialize(width, height) once, handleFrame(frame, last_frame, 0, abs_time, abs_time, frame_id) for every image in
ascending frame-id order (the first image is NOT skipped, unlike VideoProcessor, video_processor.py:167), finalize().

Export layout (image_list_processor.py:7-45): `<src_dir>/JPEGImages/<frameID>.<ext>` and `<src_dir>/JPEGImages/index.json`
= {"<frameID>": {"video_time", "frame_idx", "abs_time", "video_idx"}, ...}.

Host I/O edge, not hot path.  Decoding goes through OpenCV when it is installed and through PIL otherwise (frames are handed
on as BGR uint8, what cv2.imread returns).  Forced resolution: the reference calls cv2.resize (bilinear); without OpenCV the
resize is refused rather than approximated (deliberate refusals) -- the pipeline's configurations feed frames
at their working resolution."""
import json
import os
import time

import numpy as np


def _read_bgr(path):
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


class ImageListGenerator(object):
    """cv2.VideoCapture look-alike over the exported folder: read() -> (ok, frame); get(prop) / index2frameID() describe the frame
    read last.  Index 0 of `frameIDs` is the synthetic frame 0 the reference adds to the metadata; images start at frameIDs[1]."""

    def __init__(self, folder, extension, preload=False):
        self.folder = folder
        self.im_ext = extension[1:] if extension.startswith(".") else extension
        self.index_path = "{}/index.json".format(folder)
        with open(self.index_path, "r") as f:
            self.metadata = json.load(f)
        self.metadata["0"] = {"video_time": 0.0, "frame_idx": 0, "abs_time": 0.0, "video_idx": 0}
        self.frameIDs = sorted(int(k) for k in self.metadata)
        self.properties = self.metadata["0"].keys()
        paths = [os.path.join(folder, "{}.{}".format(fid, self.im_ext)) for fid in self.frameIDs[1:]]
        first = _read_bgr(paths[0]) if paths and os.path.exists(paths[0]) else None
        if first is None:
            raise Exception("Cannot open the file: " + (paths[0] if paths else folder))
        self.height, self.width, self.channels = first.shape
        self.preload = preload
        self.ims = np.stack([_read_bgr(p) for p in paths]) if preload else paths
        self.curr_idx = 0

    def __len__(self):
        return len(self.frameIDs) - 1

    def __getitem__(self, item):
        return self.ims[item] if self.preload else _read_bgr(self.ims[item])

    def _described(self):
        # position of the frame read last: reading image k moved curr_idx to k + 1 = its place in frameIDs; past the end the
        # reference parks the cursor at -1 (= the last frame), which is also what ends the read loop
        if self.curr_idx >= len(self):
            self.curr_idx = -1
        return self.frameIDs[self.curr_idx]

    def refresh(self):
        if self.curr_idx == -1:
            self.curr_idx = 0

    def index2frameID(self):
        return self._described()

    def read(self):
        if not 0 <= self.curr_idx < len(self):
            return False, None
        frame = self[self.curr_idx]
        self.curr_idx += 1
        return True, frame

    def get(self, prop):
        if prop not in self.properties:
            return None
        return self.metadata[str(self._described())][prop]


class ImageListProcessor:
    def __init__(self, src_dir, frames_per_second=-1, img_extension=".png"):
        self.src_dir = src_dir
        self.img_extension = img_extension
        self.frames_per_second = frames_per_second
        self.forced_width = None
        self.forced_height = None

    def force_resolution(self, width, height):
        self.forced_width = width
        self.forced_height = height

    def _resized(self, frame):
        try:
            import cv2
        except ImportError:
            raise NotImplementedError("ImageListProcessor: frames are %dx%d and %dx%d is forced; the reference resizes with cv2.resize "
                                      "(bilinear), which needs OpenCV" % (frame.shape[1], frame.shape[0], self.forced_width, self.forced_height))
        return cv2.resize(frame, (self.forced_width, self.forced_height))

    def doProcessing(self, video_worker, limit=0, verbose=False):
        if verbose:
            print("Video processing for " + video_worker.getWorkName() + " has begun")
        started = time.time()
        try:
            frames = ImageListGenerator("{}/{}".format(self.src_dir, "JPEGImages"), self.img_extension)
        except Exception as e:
            print(e)
            raise Exception("The directory <" + self.src_dir + "> is not in the correct export format, check index.json")
        forced = self.forced_width is not None
        resize = forced and (frames.width != self.forced_width or frames.height != self.forced_height)
        width, height = (self.forced_width, self.forced_height) if forced else (frames.width, frames.height)
        video_worker.initialize(width, height)
        last_frame = None
        done = 0                       # the reference's `limit` admits limit + 1 frames (its counter starts at -1)
        while limit == 0 or done <= limit:
            ok, frame = frames.read()
            if not ok:
                print("end of video reached...")
                break
            abs_time = frames.get("abs_time")
            frame_id = int(frames.index2frameID())
            if resize:
                frame = self._resized(frame)
            video_worker.handleFrame(frame, last_frame, 0, 0.0 + abs_time, abs_time, frame_id)
            if verbose and done % 50 == 0:
                print("Frames Processed = " + str(done) + ", Video Time = " + _stamp(abs_time))
            last_frame = frame
            done += 1
        video_worker.finalize()
        if verbose:
            print("Video processing for " + video_worker.getWorkName() + " completed: " + _stamp((time.time() - started) * 1000.0))


def _stamp(milliseconds):
    s, ms = divmod(int(milliseconds), 1000)
    return "%02d:%02d:%02d.%03d" % (s // 3600, s // 60 % 60, s % 60, ms)
