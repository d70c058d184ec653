"""KeyframeExtractor.GenerateFromST3DForIntervals (AccessMath/preprocessing/content/keyframe_extractor.py:13-145): one
keyframe per video segment from the space-time structure.  Same name, arguments and return values (list of H x W x 3 uint8
keyframes, ink = 0; per keyframe the sorted list of (start_time, min_x, max_x, min_y, max_y) of the groups drawn).

The reference tests ALL pairs of the segment's groups with ConnectedComponent.getOverlapFMeasure, once for the conflict
graph (CCStabilityEstimator.compute_overlapping_CC_groups, cc_stability_estimator.py:696-714) and again inside every conflict
group (:85-91); both only ask "is there a common ink pixel".  Here that question is answered once per segment on the device
(lecturemath_amd.device.image_pairs_overlap: box join + bit tests); the list / set bookkeeping that decides the drawing order
is the reference's, statement for statement, so that ties resolve the same way."""
import numpy as np

from AccessMath.data.space_time_struct import SpaceTimeStruct


class KeyframeExtractor:

    @staticmethod
    def _overlapping_groups(n_objects, pairs):
        """compute_overlapping_CC_groups (:696-748) from the list of overlapping pairs (i < j), sorted by (i, j)."""
        all_overlapping_cc = [[x] for x in range(n_objects)]
        for idx1, idx2 in pairs:                 # same append order as the reference's double loop
            all_overlapping_cc[idx1].append(idx2)
            all_overlapping_cc[idx2].append(idx1)
        group_overlap_idx = [x for x in range(n_objects)]
        merged_groups = {x: {x} for x in range(n_objects)}
        for idx in range(n_objects):
            merged_idx1 = group_overlap_idx[idx]
            for other_idx in all_overlapping_cc[idx][1:]:
                merged_idx2 = group_overlap_idx[other_idx]
                if merged_idx1 != merged_idx2:
                    merged_groups[merged_idx1] = merged_groups[merged_idx1].union(merged_groups[merged_idx2])
                    for old_group_idx in merged_groups[merged_idx2]:
                        group_overlap_idx[old_group_idx] = merged_idx1
                    del merged_groups[merged_idx2]
        overlapping_groups, no_overlaps = [], []
        for group_idx in merged_groups:
            merged_list = list(merged_groups[group_idx])
            if len(merged_list) == 1:
                no_overlaps.append(merged_list[0])
            else:
                overlapping_groups.append(merged_list)
        return overlapping_groups, no_overlaps

    @staticmethod
    def GenerateFromST3DForIntervals(st3D, video_segments, verbose=True):
        from lecturemath_amd import device
        assert isinstance(st3D, SpaceTimeStruct)
        final_keyframes = []
        keyframes_times = []
        if verbose:
            print("Total CC Groups Given: " + str(len(st3D.cc_group_boundaries)))
            print("Total Video Segments: " + str(len(video_segments)))
        for segment_idx, (start_int, end_int) in enumerate(video_segments):
            if verbose:
                print("Processing segment #{0:d} ({1:d} - {2:d})".format(segment_idx + 1, start_int, end_int))
            local_times = []
            # groups that existed in this segment, with the last of their images that overlaps the interval (:27-47)
            ids, boxes, images = [], [], []
            for group_idx in st3D.cc_group_ages:
                ages = st3D.cc_group_ages[group_idx]
                if start_int <= ages[-1] and ages[0] <= end_int:
                    last_overlap = 0
                    while last_overlap + 2 < len(ages) and ages[last_overlap + 2] <= end_int:
                        last_overlap += 1
                    ids.append(group_idx)
                    boxes.append(tuple(int(v) for v in st3D.cc_group_boundaries[group_idx]))
                    images.append(st3D.cc_group_images[group_idx][last_overlap])
            pairs = device.image_pairs_overlap(boxes, images)
            hit = set(pairs)
            overlapping_groups, no_overlaps = KeyframeExtractor._overlapping_groups(len(ids), pairs)
            frame_mask = np.zeros((st3D.height, st3D.width), dtype=np.int32)

            def draw(offset):
                min_x, max_x, min_y, max_y = boxes[offset]
                frame_mask[min_y:max_y + 1, min_x:max_x + 1] += images[offset] // 255
                start_time = st3D.frame_times[st3D.cc_group_ages[ids[offset]][0]]
                local_times.append((start_time, min_x, max_x, min_y, max_y))

            for offset in no_overlaps:
                draw(offset)
            total_in_conflict = 0
            for conflict_idx, group in enumerate(overlapping_groups):
                total_in_conflict += len(group)
                # most recent first; a group is drawn unless it shares pixels with one already accepted (:103-118)
                sorted_by_age = sorted(((st3D.cc_group_ages[ids[offset]][0], overlap_idx) for overlap_idx, offset in enumerate(group)),
                                       reverse=True)
                accepted_recent = []
                for _, overlap_idx in sorted_by_age:
                    o = group[overlap_idx]
                    if all((min(group[a], o), max(group[a], o)) not in hit for a in accepted_recent):
                        accepted_recent.append(overlap_idx)
                if verbose:
                    print("... Conflict group # {0:d}: will accept ".format(conflict_idx + 1) +
                          ",".join(str(ids[group[i]]) for i in accepted_recent))
                for overlap_idx in accepted_recent:
                    draw(group[overlap_idx])
            frame_image = np.zeros((st3D.height, st3D.width, 3), dtype=np.uint8)
            frame_image[frame_mask >= 1, :] = 255        # (:131-137: 1 -> white, >= 2 -> white as well)
            if verbose:
                print("-> Total Groups contained: " + str(len(ids)))
                print("-> Total Groups without Conflicts: " + str(len(no_overlaps)))
                print("-> Total Groups with Conflicts: " + str(total_in_conflict))
            final_keyframes.append(255 - frame_image)
            keyframes_times.append(sorted(local_times))
        return final_keyframes, keyframes_times
