"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the core of step 05 of the reference pipeline: one keyframe per video
segment from the space-time structure.  Never imported by the product; pinned against the reference itself on the G8
fixtures (tests/golden/make_golden_step05.py).  Paths relative to /root/reference/ACCESS2021_release.

  overlapping_groups   AccessMath/preprocessing/content/cc_stability_estimator.py:696-748 (compute_overlapping_CC_groups)
  keyframes            AccessMath/preprocessing/content/keyframe_extractor.py:13-145 (GenerateFromST3DForIntervals)
Pixel tests are ConnectedComponent.getOverlapFMeasure(other, False, False) (AM_CommonTools/data/connected_component.py:
202-250): recall > 0 <=> at least one common ink pixel inside the intersection of the two boxes.
"""
import numpy as np


def any_common_pixel(a, b):
    """a, b: (min_x, max_x, min_y, max_y, img uint8)."""
    ax0, ax1, ay0, ay1, ai = a
    bx0, bx1, by0, by1, bi = b
    if not (ay1 >= by0 and by1 >= ay0 and ax1 >= bx0 and bx1 >= ax0):
        return False
    x0, x1, y0, y1 = max(ax0, bx0), min(ax1, bx1), max(ay0, by0), min(ay1, by1)
    la = ai[y0 - ay0:y1 - ay0 + 1, x0 - ax0:x1 - ax0 + 1]
    lb = bi[y0 - by0:y1 - by0 + 1, x0 - bx0:x1 - bx0 + 1]
    return bool(np.count_nonzero(np.bitwise_and(la, lb)))


def overlapping_groups(objs):
    """-> (overlapping_groups, no_overlaps, hit) with the reference's merge order; hit[i][j] = pixels in common."""
    n = len(objs)
    lists = [[x] for x in range(n)]
    hit = np.zeros((n, n), bool)
    for i in range(n):
        for j in range(i + 1, n):
            if any_common_pixel(objs[i], objs[j]):
                lists[i].append(j)
                lists[j].append(i)
                hit[i, j] = hit[j, i] = True
    owner = list(range(n))
    merged = {x: {x} for x in range(n)}
    for idx in range(n):
        m1 = owner[idx]
        for other in lists[idx][1:]:
            m2 = owner[other]
            if m1 != m2:
                merged[m1] = merged[m1].union(merged[m2])
                for old in merged[m2]:
                    owner[old] = m1
                del merged[m2]
    groups, singles = [], []
    for k in merged:
        lst = list(merged[k])
        if len(lst) == 1:
            singles.append(lst[0])
        else:
            groups.append(lst)
    return groups, singles, hit


def keyframes(n_frames, width, height, frame_times, group_ages, group_bounds, group_images, segments):
    out_frames, out_times = [], []
    for start, end in segments:
        objs, ids = [], []
        for k in group_ages:
            ages = group_ages[k]
            if start <= ages[-1] and ages[0] <= end:
                last = 0
                while last + 2 < len(ages) and ages[last + 2] <= end:
                    last += 1
                mnx, mxx, mny, mxy = group_bounds[k]
                objs.append((mnx, mxx, mny, mxy, group_images[k][last]))
                ids.append(k)
        groups, singles, hit = overlapping_groups(objs)
        mask = np.zeros((height, width), np.int32)
        times = []

        def add(o):
            mnx, mxx, mny, mxy, img = objs[o]
            mask[mny:mxy + 1, mnx:mxx + 1] += img // 255
            times.append((frame_times[group_ages[ids[o]][0]], mnx, mxx, mny, mxy))

        for o in singles:
            add(o)
        for grp in groups:
            by_age = sorted(((group_ages[ids[o]][0], i) for i, o in enumerate(grp)), reverse=True)
            accepted = []
            for _, i in by_age:
                if all(not hit[grp[a], grp[i]] for a in accepted):
                    accepted.append(i)
            for i in accepted:
                add(grp[i])
        frame = np.zeros((height, width, 3), np.uint8)
        frame[mask >= 1, :] = 255
        out_frames.append(255 - frame)
        out_times.append(sorted(times))
    return out_frames, out_times
