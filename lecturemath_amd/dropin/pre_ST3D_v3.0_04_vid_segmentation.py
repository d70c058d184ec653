"""Step 04 entry point (same name, argv, config keys, inputs and output as the reference's
pre_ST3D_v3.0_04_vid_segmentation.py) for the shipped VIDEO_SEGMENTATION_METHOD = 3 (deletion events, :44-99):
[(frame_times, frame_indices, compressed_frames), (group_ages, conflicts), SpaceTimeStruct] -> list of (first, last) frame
intervals.  The reference's debug plots (matplotlib, :100-112, :175-218) and the decompression + sums that only feed them
(:28-41) are not produced; pass the parameter `sums=1` to get the binary sums printed."""
import sys
import time

import numpy as np


def deletion_signal(group_ages, st3D, add_threshold):
    """The signal method 3 segments (:44-99): every group adds its box area (as a fraction of the frame) at the frame it
    appears and at the frame it disappears; the deletions are accumulated over time, restarting from zero at every frame whose
    additions exceed add_threshold.  float64 throughout; np.add.at and np.cumsum add in index order, i.e. in the order the
    reference's loops do, so the sums round identically."""
    n = len(st3D.frame_indices)
    groups = list(group_ages)                      # dict order = group index order
    first = np.fromiter((group_ages[g][0] for g in groups), dtype=np.int64, count=len(groups))
    last = np.fromiter((group_ages[g][-1] for g in groups), dtype=np.int64, count=len(groups))
    boxes = np.array([st3D.cc_group_boundaries[g] for g in groups], dtype=np.int64).reshape(-1, 4)       # min_x, max_x, min_y, max_y
    area = ((boxes[:, 1] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 2] + 1)) / (st3D.width * st3D.height)
    added, deleted = np.zeros(n), np.zeros(n)
    np.add.at(added, first, area)
    np.add.at(deleted, last, area)
    signal = np.empty(n)
    restarts = np.flatnonzero(added > add_threshold)
    bounds = np.unique(np.concatenate([[0], restarts, [n]]))
    for a, b in zip(bounds[:-1], bounds[1:]):       # a running sum per stretch between restarts
        signal[a:b] = np.cumsum(deleted[a:b])
    return signal


def process_input(process, input_data):
    from AccessMath.data.space_time_struct import SpaceTimeStruct
    from AccessMath.preprocessing.content.video_segmenter import VideoSegmenter
    segmentation_method = process.configuration.get_int("VIDEO_SEGMENTATION_METHOD", 3)
    if segmentation_method != 3:
        raise NotImplementedError("only VIDEO_SEGMENTATION_METHOD = 3 (deletion events, the shipped configuration) is built")
    frame_times, frame_indices, compressed_frames = input_data[0]
    if "sums" in process.params:
        from AccessMath.preprocessing.content.helper import Helper
        print("Computing sums...")
        print(VideoSegmenter.compute_binary_sums(Helper.decompress_binary_images(compressed_frames)))
    group_ages, conflicts = input_data[1]
    st3D = input_data[2]
    assert isinstance(st3D, SpaceTimeStruct)
    add_threshold = process.configuration.get_float("VIDEO_SEGMENTATION_DEL_EVENT_ADD_THRESHOLD", 10)
    min_segment_length = process.configuration.get_int("VIDEO_SEGMENTATION_DEL_EVENT_MIN_LENGTH", 15)
    threshold = process.configuration.get_float("VIDEO_SEGMENTATION_DEL_EVENT_THRESHOLD", 0.25)
    cumulative_delete = deletion_signal(group_ages, st3D, add_threshold)
    n = len(cumulative_delete)
    intervals = VideoSegmenter.split_video_from_group_deletes(cumulative_delete, 0, n - 1, min_segment_length, threshold)
    print(intervals)
    print([(st3D.frame_indices[start_f], st3D.frame_indices[end_f]) for start_f, end_f in intervals])
    print("Total intervals: " + str(len(intervals)))
    return intervals


def _inputs_for_method(configuration):
    method = configuration.get_int("VIDEO_SEGMENTATION_METHOD", 2)
    keys = {3: ["CC_RECONSTRUCTED_OUTPUT", "CC_CONFLICTS_OUTPUT", "CC_ST3D_OUTPUT"], 2: ["CC_RECONSTRUCTED_OUTPUT", "CC_CONFLICTS_OUTPUT"]}
    chosen = keys.get(method)
    return [configuration.get(k) for k in chosen] if chosen else configuration.get("CC_RECONSTRUCTED_OUTPUT")


def main():
    import lm_entry
    lm_entry.run_on_inputs(sys.argv, None, "VIDEO_SEGMENTATION_OUTPUT", process_input, choose_inputs=_inputs_for_method)


if __name__ == "__main__":
    main()
