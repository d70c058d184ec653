"""Step 03 entry point (same name, argv, config keys and three outputs as the reference's pre_ST3D_v3.0_03_cc_grouping.py):
estimator -> [ (frame_times, frame_indices, clean_binary), (group_ages, conflicts), SpaceTimeStruct ]."""
import sys
import time


def process_input(process, input_data):
    from AccessMath.data.space_time_struct import SpaceTimeStruct
    frame_times, frame_indices, estimator = input_data
    cfg = process.configuration
    if "img_t" in process.params:
        img_t = float(process.params["img_t"])
    else:
        img_t = cfg.get_float("CC_GROUPING_MIN_IMAGE_THRESHOLD", 0.5)
    min_recall = cfg.get("CC_GROUPING_MIN_RECALL", 0.0)
    t_fmeasure = cfg.get("CC_GROUPING_MIN_TIME_F_MEASURE", 0.5)
    t_iou = cfg.get("CC_GROUPING_MIN_TIME_IOU", 0.25)
    max_gap = cfg.get_int("CC_STABILITY_MAX_GAP", 85)
    min_times = cfg.get_int("CC_STABILITY_MIN_TIMES", 3)
    # (the reference rebuilds the binary frames here and never uses them, :41 -- skipped)
    print("Splitting CC with large gap ... ")
    print("Total CC split: " + str(estimator.split_stable_cc_by_gaps(max_gap, min_times)))
    stable_idxs = estimator.get_stable_cc_idxs(min_times)
    print("Stable CC Count: " + str(len(stable_idxs)))
    t_window = cfg.get_int("CC_GROUPING_TEMPORAL_WINDOW", 5)
    time_ov, total, all_ov = estimator.compute_overlapping_stable_cc(stable_idxs, t_window)
    print("Total intersections found: " + str(total))
    cc_groups, group_idx_per_cc = estimator.compute_groups(stable_idxs, time_ov, min_recall, t_fmeasure, t_iou)
    print("Final count of groups: " + str(len(cc_groups)))
    group_ages, groups_per_frame = estimator.compute_groups_temporal_information(cc_groups)
    conflicts = estimator.compute_conflicting_groups(stable_idxs, all_ov, len(cc_groups), group_idx_per_cc)
    group_images, group_boundaries = estimator.compute_group_images(cc_groups, group_ages, img_t)
    clean_binary = estimator.frames_from_groups(cc_groups, group_boundaries, groups_per_frame, group_ages, group_images, None,
                                                min_times, True)
    st3d = SpaceTimeStruct(frame_times, frame_indices, estimator.height, estimator.width, group_ages, group_images, group_boundaries)
    return [(frame_times, frame_indices, clean_binary), (group_ages, conflicts), st3d]


def main():
    import lm_entry
    lm_entry.run_on_inputs(sys.argv, "CC_STABILITY_OUTPUT", ["CC_RECONSTRUCTED_OUTPUT", "CC_CONFLICTS_OUTPUT", "CC_ST3D_OUTPUT"], process_input)


if __name__ == "__main__":
    main()
