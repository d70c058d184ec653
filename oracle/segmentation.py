"""TEST INFRASTRUCTURE ONLY -- CPU restatement of step 04 of the reference pipeline, deletion-event video segmentation
(VIDEO_SEGMENTATION_METHOD = 3).  Never imported by the product; pinned against the reference itself on the G7 fixtures
(tests/golden/make_golden_step04.py).  Paths relative to /root/reference/ACCESS2021_release.

  deletion_signals   pre_ST3D_v3.0_04_vid_segmentation.py:56-87   add / delete / cumulative-delete signals from group ages + boxes
  find_signal_peaks  AccessMath/preprocessing/content/video_segmenter.py:133-182
  split              video_segmenter.py:499-520 (recursive split at the highest admissible peak)
  binary_sums        video_segmenter.py:22-28
"""
import numpy as np


def binary_sums(frames):
    """sum of every {0,255} frame divided by 255 (python float division of a numpy integer sum)."""
    return [f.sum() / 255 for f in frames]


def deletion_signals(n_frames, width, height, group_ages, group_bounds, add_threshold):
    add_values = np.zeros(n_frames)
    del_values = np.zeros(n_frames)
    for k in group_ages:                        # dict order == group index order
        first, last = group_ages[k][0], group_ages[k][-1]
        mnx, mxx, mny, mxy = group_bounds[k]
        area = (mxx - mnx + 1) * (mxy - mny + 1)
        area /= (width * height)                # true division, float64
        add_values[first] += area
        del_values[last] += area
    cumulative = np.zeros(n_frames)
    acc = 0.0
    for i in range(n_frames):
        if add_values[i] > add_threshold:
            acc = 0.0
        acc += del_values[i]
        cumulative[i] = acc
    return add_values, del_values, cumulative


def find_signal_peaks(start, end, signal):
    peaks = []
    p_start = p_high = None
    going_up = None
    for i in range(start, end + 1):
        if p_start is None:
            p_start = p_high = i
            going_up = True
        elif signal[i] > signal[i - 1]:
            if going_up:
                p_high = i
            else:
                peaks.append((p_start, p_high, i - 1))
                p_start = p_high = i
                going_up = True
        elif signal[i] < signal[i - 1]:
            going_up = False
    if p_start is not None:
        peaks.append((p_start, p_high, end))
    return peaks


def split(signal, start, end, min_length, threshold):
    cands = []
    for _, high, _ in find_signal_peaks(start, end, signal):
        if signal[high] > threshold and start + min_length <= high <= end - min_length:
            cands.append((signal[high], high))
    cands.sort(reverse=True)
    if not cands:
        return [(start, end)]
    best = cands[0][1]
    return split(signal, start, best - 1, min_length, threshold) + split(signal, best + 1, end, min_length, threshold)


def run_step04(n_frames, width, height, group_ages, group_bounds, add_threshold=10, min_length=15, threshold=0.25):
    _, _, cumulative = deletion_signals(n_frames, width, height, group_ages, group_bounds, add_threshold)
    return split(cumulative, 0, n_frames - 1, min_length, threshold)
