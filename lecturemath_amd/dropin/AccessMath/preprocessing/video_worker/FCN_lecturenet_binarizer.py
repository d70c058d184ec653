"""Step-01 video worker with the reference's protocol (video_worker/FCN_lecturenet_binarizer.py:30-79):
initialize(w, h) / handleFrame(frame, last_frame, v_index, abs_time, rel_time, abs_frame_idx) / getWorkName() / finalize(),
results in frame_times, frame_indices, compressed_frames (PNG byte arrays).  No OpenCV: BGR->RGB is a slice, PNG is zlib."""
import PIL.Image

from lecturemath_amd import png


class FCN_LectureNet_Binarizer:
    def __init__(self, lecture_net):
        self.width = self.height = 0
        self.frame_count = 0
        self.lecture_net = lecture_net
        self.last_binary = self.last_text = self.last_rec = None
        self.frame_times = self.frame_indices = self.compressed_frames = None
        self.debug_mode = False
        self.debug_start = self.debug_end = 0.0
        self.debug_out_dir = None
        self.debug_video_name = ""

    def initialize(self, width, height):
        self.width, self.height = width, height
        self.frame_count = 0
        self.frame_times, self.frame_indices, self.compressed_frames = [], [], []

    def set_debug_mode(self, active, start_time, end_time, out_dir, video_name):
        self.debug_mode, self.debug_start, self.debug_end = active, start_time, end_time
        self.debug_out_dir, self.debug_video_name = out_dir, video_name

    def handleFrame(self, frame, last_frame, v_index, abs_time, rel_time, abs_frame_idx):
        self.frame_count += 1
        pil_image = PIL.Image.fromarray(frame[:, :, ::-1].copy())          # BGR -> RGB
        binary, text_mask, rec_img = self.lecture_net.binarize(pil_image, return_others=True, force_binary=True)
        binary = 255 - binary                                              # ink = 255 from here on
        self.last_binary, self.last_text, self.last_rec = binary, text_mask, rec_img
        self.compressed_frames.append(png.encode_gray8(binary))
        self.frame_indices.append(abs_frame_idx)
        self.frame_times.append(abs_time)
        if self.debug_mode and self.debug_start <= abs_time <= self.debug_end:
            self.debug_frame(binary)

    def debug_frame(self, binary):
        name = self.debug_out_dir + "/binary_" + self.debug_video_name + "_" + str(self.frame_count) + ".png"
        with open(name, "wb") as f:
            f.write(png.encode_gray8(binary).tobytes())

    def getWorkName(self):
        return "FCN_LectureNet Frame Binarizer"

    def finalize(self):
        pass
