"""KeyframeExtractor.GenerateFromST3DForIntervals (AccessMath/preprocessing/content/keyframe_extractor.py:13-145): one
keyframe per video segment from the space-time structure.  Same name, arguments and return values (list of H x W x 3 uint8
keyframes, ink = 0; per keyframe the sorted list of (start_time, min_x, max_x, min_y, max_y) of the groups drawn).

What the reference computes per segment, restated:
  1. the groups alive in the segment and, for each, the last of its images that overlaps the segment (:27-47);
  2. the connected components of the "shares an ink pixel" graph over those groups
     (CCStabilityEstimator.compute_overlapping_CC_groups, cc_stability_estimator.py:696-748) -- every pair is tested with
     ConnectedComponent.getOverlapFMeasure there, and AGAIN inside every component (:85-91);
  3. isolated groups are drawn; inside a component the groups are visited from the most recently started to the oldest and a
     group is drawn unless it shares pixels with one drawn before it (:103-118).
Here the pixel question of 2. and 3. is answered ONCE per segment on the device (lecturemath_amd.device.image_pairs_overlap:
box join + bit tests), and the rest works on index arrays.

Order contract.  Only the SET of groups drawn reaches the outputs (the mask is a sum, the times are sorted), and that set
depends on the visiting order of step 3: descending (first frame of the group, position of the group inside its component).
The reference takes the position from `list(set)` of a component that was assembled by `set.union` calls -- CPython's set
iteration order for small ints, which is NOT ascending once values exceed the table size ({5, 300} iterates as 300, 5) and
depends on the sequence of unions.  Groups of one component that start at the same frame and overlap are therefore resolved by
that order; `_component_member_order` reproduces it by performing the same unions on real Python sets, in the sequence the
reference's scan produces (ascending object, ascending partner), and nothing else of it."""
import numpy as np

from AccessMath.data.space_time_struct import SpaceTimeStruct


class KeyframeExtractor:

    @staticmethod
    def _component_member_order(n_objects, pairs):
        """Connected components of the overlap graph.  pairs: (i < j), sorted.  Returns (components with >= 2 members, each as
        the member list in the reference's order; isolated objects), both in ascending order of their smallest member."""
        pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        # symmetric adjacency in CSR form, partners ascending (the order in which the reference's double loop appends them)
        both = np.concatenate([pairs, pairs[:, ::-1]])
        both = both[np.lexsort((both[:, 1], both[:, 0]))]
        start = np.searchsorted(both[:, 0], np.arange(n_objects + 1))
        owner = list(range(n_objects))                      # representative (= smallest member) of every object's component
        members = {k: {k} for k in range(n_objects)}        # representative -> CPython set of members (the order carrier)
        for obj in range(n_objects):
            mine = owner[obj]
            for other in both[start[obj]:start[obj + 1], 1].tolist():
                theirs = owner[other]
                if theirs == mine:
                    continue
                absorbed = members.pop(theirs)
                members[mine] = members[mine].union(absorbed)
                for k in absorbed:
                    owner[k] = mine
        components = [list(v) for v in members.values() if len(v) > 1]
        isolated = [next(iter(v)) for v in members.values() if len(v) == 1]
        return components, isolated

    @staticmethod
    def GenerateFromST3DForIntervals(st3D, video_segments, verbose=True):
        from lecturemath_amd import device
        assert isinstance(st3D, SpaceTimeStruct)
        group_ids = list(st3D.cc_group_ages)
        first = np.array([st3D.cc_group_ages[g][0] for g in group_ids], dtype=np.int64)
        last = np.array([st3D.cc_group_ages[g][-1] for g in group_ids], dtype=np.int64)
        keyframes, keyframe_times = [], []
        if verbose:
            print("%d CC groups, %d video segments" % (len(group_ids), len(video_segments)))
        for seg_no, (seg_first, seg_last) in enumerate(video_segments):
            alive = np.flatnonzero((seg_first <= last) & (first <= seg_last))
            ids = [group_ids[k] for k in alive]
            boxes, images = [], []
            for g in ids:
                ages = st3D.cc_group_ages[g]
                # segment image k covers [ages[k], ages[k + 1]]: the last one whose end lies inside the video segment, else the first
                k = max(0, int(np.searchsorted(ages, seg_last, side="right")) - 2)
                boxes.append(tuple(int(v) for v in st3D.cc_group_boundaries[g]))
                images.append(st3D.cc_group_images[g][k])
            pairs = device.image_pairs_overlap(boxes, images)
            n = len(ids)
            clash = np.zeros((n, n), dtype=bool)
            if pairs:
                pa = np.asarray(pairs, dtype=np.int64)
                clash[pa[:, 0], pa[:, 1]] = clash[pa[:, 1], pa[:, 0]] = True
            components, isolated = KeyframeExtractor._component_member_order(n, pairs)
            drawn = list(isolated)
            for comp in components:
                # most recently started first; equal starts: later position in the component first
                order = sorted(range(len(comp)), key=lambda pos: (first[alive[comp[pos]]], pos), reverse=True)
                kept = []
                for pos in order:
                    if not clash[comp[pos], kept].any():
                        kept.append(comp[pos])
                drawn.extend(kept)
                if verbose:
                    print("  segment %d: %d overlapping groups, keeping %s" % (seg_no + 1, len(comp), ",".join(str(ids[o]) for o in kept)))
            mask = np.zeros((st3D.height, st3D.width), dtype=bool)
            times = []
            for o in drawn:
                x0, x1, y0, y1 = boxes[o]
                mask[y0:y1 + 1, x0:x1 + 1] |= images[o] > 0
                times.append((st3D.frame_times[first[alive[o]]], x0, x1, y0, y1))
            frame = np.full((st3D.height, st3D.width, 3), 255, dtype=np.uint8)      # white board, ink = 0 (:131-141)
            frame[mask] = 0
            if verbose:
                print("  segment %d (%d - %d): %d groups, %d isolated, %d in %d overlapping sets" %
                      (seg_no + 1, seg_first, seg_last, n, len(isolated), sum(len(c) for c in components), len(components)))
            keyframes.append(frame)
            keyframe_times.append(sorted(times))
        return keyframes, keyframe_times
