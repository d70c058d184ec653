"""VideoSegmenter (AccessMath/preprocessing/content/video_segmenter.py) -- the parts step 04 uses with the shipped
configuration (VIDEO_SEGMENTATION_METHOD = 3, deletion events): compute_binary_sums (:22-28), find_signal_peaks (:133-182),
split_video_from_group_deletes (:499-520).  Same names, arguments and return values.

compute_binary_sums accepts what the reference accepts (a list of uint8 frames) and, additionally, a device tensor [n, H, W]
as produced by CCStabilityEstimator.frames_from_groups_device -- then the sums are reduced on the GPU (lm_frame_sums).
The sklearn decision-tree method (1) and the conflict-minimisation method (2) are not part of this build."""
import numpy as np


class VideoSegmenter:
    ConflictsAreaWeightsCount = 0
    ConflictsAreaWeigthsUnion = 3
    ConflictsAreaWeightsIntersection = 4
    ConflictsAreaWeightsIOU = 5

    ConflictsPixelsWeightsNone = 0
    ConflictsPixelsWeightsMatched = 1
    ConflictsPixelsWeightsUnmatched = 2
    ConflictsPixelsWeightsIOU = 3

    ConflictsTimeWeightNone = 0
    ConflictsTimeWeightGap = 1
    ConflictsTimeWeightNormalizedLength = 2

    @staticmethod
    def compute_binary_sums(all_binary):
        if not isinstance(all_binary, (list, tuple)) and hasattr(all_binary, "shape") and len(all_binary.shape) == 3 and \
                not isinstance(all_binary, np.ndarray):
            from lecturemath_amd import device
            return [int(v) / 255 for v in device.frame_sums(all_binary)]
        return [binary.sum() / 255 for binary in all_binary]

    @staticmethod
    def find_signal_peaks(start_frame, end_frame, signal_dict):
        all_peaks = []
        peak_start = peak_highest = None
        going_up = None
        for frame_idx in range(start_frame, end_frame + 1):
            if peak_start is None:
                peak_start = peak_highest = frame_idx
                going_up = True
            elif signal_dict[frame_idx] > signal_dict[frame_idx - 1]:
                if going_up:
                    peak_highest = frame_idx
                else:                                   # going down and now up again: the peak ends, a new one starts
                    all_peaks.append((peak_start, peak_highest, frame_idx - 1))
                    peak_start = peak_highest = frame_idx
                    going_up = True
            elif signal_dict[frame_idx] < signal_dict[frame_idx - 1]:
                going_up = False
        if peak_start is not None:
            all_peaks.append((peak_start, peak_highest, end_frame))
        return all_peaks

    @staticmethod
    def split_video_from_group_deletes(signal, start_frame, end_frame, min_length, threshold):
        candidate_peaks = []
        for _, peak_highest, _ in VideoSegmenter.find_signal_peaks(start_frame, end_frame, signal):
            if signal[peak_highest] > threshold and start_frame + min_length <= peak_highest <= end_frame - min_length:
                candidate_peaks.append((signal[peak_highest], peak_highest))
        candidate_peaks = sorted(candidate_peaks, reverse=True)
        if len(candidate_peaks) == 0:
            print(str([(start_frame, end_frame)]) + " no good split candidates found")
            return [(start_frame, end_frame)]
        _, best_split = candidate_peaks[0]
        left = VideoSegmenter.split_video_from_group_deletes(signal, start_frame, best_split - 1, min_length, threshold)
        right = VideoSegmenter.split_video_from_group_deletes(signal, best_split + 1, end_frame, min_length, threshold)
        return left + right

    @staticmethod
    def video_segments_from_sums(all_sums, leaf_min, min_erase_ratio):
        raise NotImplementedError("VIDEO_SEGMENTATION_METHOD 1 (sums + decision tree) is not part of this build; use method 3")

    @staticmethod
    def from_group_conflicts(*args, **kwargs):
        raise NotImplementedError("VIDEO_SEGMENTATION_METHOD 2 (conflict minimisation) is not part of this build; use method 3")
