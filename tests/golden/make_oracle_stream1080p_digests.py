#!/usr/bin/env python3
"""Digests of the FULL 10,000-frame 1080p stream (BASELINE configs[2]) from the ORACLE (oracle/cc_oracle.c for step 02,
oracle/grouping.py for the step-03 tables) -- because the reference itself cannot process this stream in the build container:
it keeps every CC object of every frame with its uint8 mask plus ~19 GB of group images, and was killed by the OOM killer at
65 GB RSS (64 GB box, no swap) in compute_group_images.  The oracle is pinned to the reference on the golden fixtures and on the
first 1,000 frames of this very stream (g9_stream1080p_digests.json["1000"], produced by the reference).

    python tests/golden/make_oracle_stream1080p_digests.py [n_frames]        # default 10000; tens of minutes, ~30 GB
    python tests/golden/make_oracle_stream1080p_digests.py --4k 1024         # configs[4]: 3840 x 2160 -> g9_stream4k_digests.json

Writes g9_stream1080p_digests.json["<n>"] with "produced_by": "oracle"; group images and reconstructed frames are not
produced (the oracle's numpy version of them needs the same tens of GB)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from lecturemath_amd import digests, synth  # noqa: E402
from oracle import cc as occ  # noqa: E402
from oracle import grouping as og  # noqa: E402

OUT = os.path.join(HERE, "g9_stream1080p_digests.json")
H, W, SEED = 1080, 1920, 20213
if "--4k" in sys.argv:          # BASELINE configs[4]: the same generator at 3840 x 2160 (bench.py --height 2160 --width 3840 --frames N)
    sys.argv.remove("--4k")
    OUT = os.path.join(HERE, "g9_stream4k_digests.json")
    H, W = 2160, 3840


def run(n):
    t0 = time.time()
    st = occ.Stability(W, H, 0.85, 0.85, 85)
    for i, f in enumerate(synth.binary_stream(n, H, W, seed=SEED)):
        st.add_frame(f)
        if i % 1000 == 999:
            print("  step 02: frame %d, %.0f s" % (i + 1, time.time() - t0), flush=True)
    state = st.result()
    t1 = time.time()
    d = {"produced_by": "oracle", "tempo_count": int(state["tempo_count"]), "n_unique_step02": len(state["unique_cc_frames"]),
         "n_cc": int(sum(len(fr) for fr in state["cc_idx_per_frame"])),
         "unique_cc_frames_step02": digests.lists_digest(state["unique_cc_frames"], 2),
         "cc_idx_per_frame_step02": digests.lists_digest(state["cc_idx_per_frame"], 2)}
    print("  step 02 done: %d uniques, %.0f s" % (d["n_unique_step02"], t1 - t0), flush=True)
    n_split = og.split_stable_cc_by_gaps(state, 85, 3)
    stable = og.stable_idxs(state, 3)
    time_ov, total, all_ov = og.overlapping_stable_cc(state, stable, 5)
    print("  overlaps done: %d stable, %d intersections, %.0f s" % (len(stable), total, time.time() - t1), flush=True)
    groups, gid = og.compute_groups(stable, time_ov, 0.5)
    ages, per_frame = og.groups_temporal_information(state, groups)
    recs = state["unique_recs"]
    bounds = {}
    for g, members in enumerate(groups):
        r = recs[np.asarray(members)]
        bounds[g] = (int(r[:, 0].min()), int(r[:, 1].max()), int(r[:, 2].min()), int(r[:, 3].max()))
    d.update(digests.from_python(state["unique_cc_frames"], state["cc_idx_per_frame"], groups, ages, per_frame, bounds))
    d.update({"n_split": int(n_split), "n_stable": len(stable), "n_groups": len(groups), "total_intersections": int(total),
              "n_unique": len(state["unique_cc_frames"]), "n_frames": n,
              "oracle_seconds": {"step02": round(t1 - t0, 1), "step03_tables": round(time.time() - t1, 1)}})
    return d


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    d = run(n)
    allv = json.load(open(OUT)) if os.path.exists(OUT) else {}
    key = str(n) if str(n) not in allv or allv[str(n)].get("produced_by") == "oracle" else str(n) + "_oracle"
    allv[key] = d
    json.dump(allv, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(d, indent=1), flush=True)
