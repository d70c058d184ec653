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
        """Peaks of signal[start_frame .. end_frame] as (first frame, frame of the maximum reached while rising, last frame)
        (video_segmenter.py:133-182).  A peak ends where the signal, having fallen, rises again; plateaus keep the direction.
        Formulated on the sign of the first difference: the peaks are delimited by the rising steps that follow a falling
        step, and a peak's top is the last rising step before its first falling one."""
        n = end_frame - start_frame + 1
        if n <= 0:
            return []
        values = np.fromiter((signal_dict[f] for f in range(start_frame, end_frame + 1)), dtype=np.float64, count=n)
        step = np.sign(np.diff(values)).astype(np.int8)             # step[k]: frame start + k + 1 against the one before
        moves = np.flatnonzero(step)                                 # plateaus carry no information
        rising = step[moves] > 0
        # a new peak starts at every rising step whose previous move was a fall
        starts_new = np.flatnonzero(rising[1:] & ~rising[:-1]) + 1 if len(moves) > 1 else np.zeros(0, np.int64)
        first_move = np.concatenate([[0], starts_new])              # index into `moves` of every peak's first move
        last_move = np.concatenate([starts_new, [len(moves)]])      # one past its last move
        begins = np.concatenate([[start_frame], start_frame + 1 + moves[starts_new]]) if len(moves) else np.array([start_frame])
        ends = np.concatenate([begins[1:] - 1, [end_frame]])
        peaks = []
        for k in range(len(begins)):
            top = int(begins[k])
            seg = rising[first_move[k]:last_move[k]] if len(moves) else rising[:0]
            if len(seg) and seg[0]:                                  # leading run of rises (the very first peak may start falling)
                falls = np.flatnonzero(~seg)
                run = len(seg) if len(falls) == 0 else int(falls[0])
                top = start_frame + 1 + int(moves[first_move[k] + run - 1])
            peaks.append((int(begins[k]), top, int(ends[k])))
        return peaks

    @staticmethod
    def split_video_from_group_deletes(signal, start_frame, end_frame, min_length, threshold):
        """Recursive split at the highest sufficiently prominent peak top that leaves min_length frames on both sides
        (video_segmenter.py:499-520); ties go to the later frame.  Returned intervals are in temporal order.  Worked through
        with an explicit stack (right part pushed first, so leaves come out left to right like the recursion's)."""
        intervals = []
        todo = [(start_frame, end_frame)]
        while todo:
            lo, hi = todo.pop()
            tops = np.array([top for _, top, _ in VideoSegmenter.find_signal_peaks(lo, hi, signal)], dtype=np.int64)
            if len(tops):
                heights = np.array([signal[t] for t in tops], dtype=np.float64)
                keep = (heights > threshold) & (tops >= lo + min_length) & (tops <= hi - min_length)
                tops, heights = tops[keep], heights[keep]
            if len(tops) == 0:
                print(str([(lo, hi)]) + " no good split candidates found")
                intervals.append((lo, hi))
                continue
            best = int(tops[np.lexsort((tops, heights))[-1]])       # highest, then latest
            todo.append((best + 1, hi))
            todo.append((lo, best - 1))
        return intervals

    @staticmethod
    def video_segments_from_sums(all_sums, leaf_min, min_erase_ratio):
        raise NotImplementedError("VIDEO_SEGMENTATION_METHOD 1 (sums + decision tree) is not part of this build; use method 3")

    @staticmethod
    def from_group_conflicts(*args, **kwargs):
        raise NotImplementedError("VIDEO_SEGMENTATION_METHOD 2 (conflict minimisation) is not part of this build; use method 3")
