"""Step-03 (CC grouping) CPU oracle over plain data -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, with numpy and ordinary Python containers, what the reference's step 03 computes from a
finished CCStabilityEstimator.  Paths relative to /root/reference/ACCESS2021_release; every function
names the method it follows in AccessMath/preprocessing/content/cc_stability_estimator.py.

Parity status: pinned against the reference run in the build container
(tests/golden/make_golden.py G4 fixtures; tests/test_oracle_golden.py, tests/test_reference_binding.py).

Input `state` is the dict produced by oracle.cc.Stability.result():
    unique_recs      int32 [U,5]  (min_x, max_x, min_y, max_y, size) of each unique CC's first-seen mask
    unique_crops     list of uint8 0/255 arrays
    unique_cc_frames list of [(frame, raw_label), ...]
    cc_idx_per_frame list of [(unique_idx, cc_id), ...]
All functions treat `state` as mutable in the same places the reference mutates its estimator.
"""
import numpy as np

from . import cc as _cc


def split_stable_cc_by_gaps(state, max_gap, stable_min_frames):
    """split_stable_cc_by_gaps, cc_stability_estimator.py:181-228."""
    recs = [tuple(int(v) for v in r) for r in state["unique_recs"]]
    crops = state["unique_crops"]
    frames = state["unique_cc_frames"]
    per_frame = state["cc_idx_per_frame"]
    n_split = 0
    for u in range(len(frames)):
        fl = frames[u]
        runs = [[fl[0]]]
        for prev, cur in zip(fl[:-1], fl[1:]):
            if cur[0] - prev[0] > max_gap:
                runs.append([cur])
            else:
                runs[-1].append(cur)
        if len(runs) < 2 or len(fl) < stable_min_frames:
            continue
        frames[u] = runs[0]
        for later in runs[1:]:
            new_u = len(frames)
            recs.append(recs[u])
            crops.append(crops[u])
            frames.append(later)
            for f_idx, _lbl in later:
                entries = per_frame[f_idx]
                for k, (uu, cid) in enumerate(entries):
                    if uu == u:            # first entry of that unique in the frame only
                        entries[k] = (new_u, cid)
                        break
        n_split += 1
    state["unique_recs"] = np.asarray(recs, dtype=np.int32).reshape(-1, 5)
    return n_split


def stable_idxs(state, min_frames):
    """get_stable_cc_idxs, :230-236."""
    return [u for u, fl in enumerate(state["unique_cc_frames"]) if len(fl) >= min_frames]


def _bbox_pairs(recs, idxs, chunk=2048):
    """All (a, b), a < b, among idxs whose inclusive boxes intersect; sorted.  Equivalent to the
    IntervalIndex X-join intersected with the Y-join at :255-272 (tools/interval_index.py:42-99)."""
    idxs = np.asarray(idxs, dtype=np.int64)
    if len(idxs) == 0:
        return []
    b = recs[idxs].astype(np.int64)
    out = []
    for s in range(0, len(idxs), chunk):
        a = b[s:s + chunk]
        hit = ((a[:, None, 0] <= b[None, :, 1]) & (b[None, :, 0] <= a[:, None, 1]) &
               (a[:, None, 2] <= b[None, :, 3]) & (b[None, :, 2] <= a[:, None, 3]))
        ii, jj = np.nonzero(hit)
        ga, gb = idxs[s + ii], idxs[jj]
        keep = ga < gb
        out.append(np.stack([ga[keep], gb[keep]], axis=1))
    pairs = np.concatenate(out, axis=0)
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    return [(int(a), int(c)) for a, c in pairs[order]]


def overlapping_stable_cc(state, stable, temporal_window):
    """compute_overlapping_stable_cc, :245-306.  Returns (time_overlapping, total, all_overlapping)."""
    recs, crops, frames = state["unique_recs"], state["unique_crops"], state["unique_cc_frames"]
    n = len(frames)
    all_ov = [[] for _ in range(n)]
    time_ov = [[] for _ in range(n)]
    total = 0
    for a, b in _bbox_pairs(recs, stable):
        match = _cc.overlap(recs[a, :4], crops[a], recs[b, :4], crops[b])
        size_a, size_b = np.int32(recs[a, 4]), np.int32(recs[b, 4])
        recall = match / float(size_a)          # connected_component.py:239
        precision = match / float(size_b)       # :240
        if recall > 0.0 or precision > 0.0:
            matched_pixels = int(size_a * recall)    # float64 round trip, can be match-1 (:294)
            all_ov[a].append((b, matched_pixels, int(size_b), int(size_a)))
            all_ov[b].append((a, matched_pixels, int(size_a), int(size_b)))
            a0, a1 = frames[a][0][0], frames[a][-1][0]
            b0, b1 = frames[b][0][0], frames[b][-1][0]
            if a1 + temporal_window >= b0 and b1 >= a0 - temporal_window:
                time_ov[a].append((b, recall, precision))
                time_ov[b].append((a, precision, recall))
                total += 1
    return time_ov, total, all_ov


def compute_groups(stable, time_ov, min_recall):
    """compute_groups, :308-413 (the t_fmeasure / t_time_IOU arguments are dead code there)."""
    groups = []
    gid = {}
    for a in stable:
        if a in gid:
            g = gid[a]
        else:
            g = len(groups)
            groups.append([a])
            gid[a] = g
        for b, recall, _precision in time_ov[a]:
            if recall < min_recall:
                continue
            if b not in gid:
                gid[b] = g
                groups[g].append(b)
            else:
                og = gid[b]
                if og != g:
                    for m in groups[og]:
                        gid[m] = g
                        groups[g].append(m)
                    groups[og] = []
    final, final_gid = [], {}
    for grp in groups:
        if grp:
            k = len(final)
            final.append(grp)
            for m in grp:
                final_gid[m] = k
    return final, final_gid


def groups_temporal_information(state, groups):
    """compute_groups_temporal_information, :415-444."""
    frames = state["unique_cc_frames"]
    n_frames = len(state["cc_idx_per_frame"])
    ages = {}
    per_frame = [[] for _ in range(n_frames)]
    for g, members in enumerate(groups):
        if not members:
            continue
        seen = []
        for u in members:
            for t in (frames[u][0][0], frames[u][-1][0]):
                if t not in seen:
                    seen.append(t)
        seen.sort()
        ages[g] = seen
        for f in range(seen[0], min(seen[-1] + 1, n_frames)):
            per_frame[f].append(g)
    return ages, per_frame


def _box_area(r):
    return (int(r[1]) - int(r[0]) + 1) * (int(r[3]) - int(r[2]) + 1)


def _box_overlap_area(a, b):
    """ConnectedComponent.getOverlapArea, AM_CommonTools/data/connected_component.py:54-67 (float 0.0 if disjoint)."""
    if a[0] <= b[1] and b[0] <= a[1] and a[2] <= b[3] and b[2] <= a[3]:
        return (min(int(a[1]), int(b[1])) - max(int(a[0]), int(b[0])) + 1) * \
               (min(int(a[3]), int(b[3])) - max(int(a[2]), int(b[2])) + 1)
    return 0.0


def conflicting_groups(state, stable, all_ov, n_groups, gid):
    """compute_conflicting_groups, :446-500."""
    recs = state["unique_recs"]
    conflicts = {g: {} for g in range(n_groups)}
    for a in stable:
        area_a = _box_area(recs[a])
        for b, matched, size_b, size_a in all_ov[a]:
            if not a < b:
                continue
            unmatched = size_a + size_b - matched * 2
            inter = _box_overlap_area(recs[a], recs[b])
            union = area_a + _box_area(recs[b]) - inter
            ga, gb = gid[a], gid[b]
            if ga == gb:
                continue
            for x, y in ((ga, gb), (gb, ga)):
                d = conflicts[x].setdefault(y, {"matched": 0, "unmatched": 0, "area_union": 0, "area_intersection": 0})
                d["matched"] += matched
                d["unmatched"] += unmatched
                d["area_union"] += union
                d["area_intersection"] += inter
    return conflicts


def group_images(state, groups, ages, threshold):
    """compute_group_images, :575-636."""
    recs, crops, frames = state["unique_recs"], state["unique_crops"], state["unique_cc_frames"]
    images, bounds = {}, {}
    for g, members in enumerate(groups):
        if not members:
            continue
        r = recs[members]
        x0, x1, y0, y1 = int(r[:, 0].min()), int(r[:, 1].max()), int(r[:, 2].min()), int(r[:, 3].max())
        bounds[g] = (x0, x1, y0, y1)
        segs = []
        for t0, t1 in zip(ages[g][:-1], ages[g][1:]):
            acc = np.zeros((y1 - y0 + 1, x1 - x0 + 1), np.int32)
            for u in members:
                times = sum(1 for f, _ in frames[u] if t0 <= f <= t1)   # duplicates count (:619)
                if times:
                    ux0, uy0 = int(recs[u, 0]) - x0, int(recs[u, 2]) - y0
                    c = crops[u]
                    acc[uy0:uy0 + c.shape[0], ux0:ux0 + c.shape[1]] += (c // 255).astype(np.int32) * times
            with np.errstate(divide="ignore", invalid="ignore"):
                seg = ((acc.astype(np.float64) / acc.max()) >= threshold).astype(np.uint8) * 255
            segs.append(seg)
        images[g] = segs
    return images, bounds


def frames_from_groups(state, groups, bounds, per_frame, ages, images):
    """frames_from_groups(save_prefix=None, show_unstable=True), :638-681 -- channel 0 only, which is the
    only channel that is encoded (:678); uint8 adds wrap (:660)."""
    h = int(state["height"])
    w = int(state["width"])
    seg_ptr = [0] * len(groups)
    out = []
    for t, live in enumerate(per_frame):
        canvas = np.zeros((h, w), np.uint8)
        for g in live:
            a = ages[g]
            while a[seg_ptr[g] + 1] < t:
                seg_ptr[g] += 1
            x0, x1, y0, y1 = bounds[g]
            canvas[y0:y1 + 1, x0:x1 + 1] += images[g][seg_ptr[g]]
        out.append(canvas)
    return out


def run_step03(state, max_gap=85, min_times=3, t_window=5, min_recall=0.5, img_threshold=0.5, reconstruct=True):
    """process_input of pre_ST3D_v3.0_03_cc_grouping.py:22-118 (without prints / the dead rebuilt_binary_images)."""
    n_split = split_stable_cc_by_gaps(state, max_gap, min_times)
    stable = stable_idxs(state, min_times)
    time_ov, total, all_ov = overlapping_stable_cc(state, stable, t_window)
    groups, gid = compute_groups(stable, time_ov, min_recall)
    ages, per_frame = groups_temporal_information(state, groups)
    conflicts = conflicting_groups(state, stable, all_ov, len(groups), gid)
    images, bounds = group_images(state, groups, ages, img_threshold)
    clean = frames_from_groups(state, groups, bounds, per_frame, ages, images) if reconstruct else None
    return {
        "n_split": n_split, "stable_idxs": stable, "time_overlapping_cc": time_ov, "total_intersections": total,
        "all_overlapping_cc": all_ov, "cc_groups": groups, "group_idx_per_cc": gid, "group_ages": ages,
        "groups_per_frame": per_frame, "conflicts": conflicts, "group_images": images,
        "group_boundaries": bounds, "clean_binary": clean,
    }
