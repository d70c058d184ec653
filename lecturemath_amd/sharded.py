"""Frame-range sharding of ONE stream across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The per-frame stages (threshold, labelling, statistics, CC records + crops; labeler.py:116-191) are independent per frame and
run on the rank that owns the frame range; temporal matching (cc_stability_estimator.py:71-145) carries state from frame to
frame and first-match-wins against FIRST-SEEN masks, so it is replayed sequentially on rank 0 over the gathered records --
KBs per frame.  The only collectives are that gather and the one-off broadcast of the FCN weights.
Independent lectures need no collective at all (bench.py --gpus N).
"""
import numpy as np

from . import device


def frame_range(n_frames, rank, world):
    """Contiguous block of ceil(F/R) frames per rank."""
    per = -(-n_frames // world)
    return min(rank * per, n_frames), min((rank + 1) * per, n_frames)


def broadcast_state_dict(sd, src=0, device_name=None):
    """Rank `src` holds the FCN state_dict (147 tensors, 63 MB fp32 at the shipped widths); every rank gets a copy."""
    import torch
    import torch.distributed as dist
    keys = [sorted(sd.keys())] if dist.get_rank() == src else [None]
    dist.broadcast_object_list(keys, src=src)
    meta = [[(k, tuple(sd[k].shape), str(sd[k].dtype).replace("torch.", "")) for k in keys[0]]] if dist.get_rank() == src else [None]
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for k, shape, dt in meta[0]:
        dtype = getattr(torch, dt)
        if dist.get_rank() == src:
            t = sd[k].to(device_name) if device_name else sd[k].clone()
        else:
            t = torch.empty(shape, dtype=dtype, device=device_name or "cpu")
        dist.broadcast(t.contiguous(), src=src)
        out[k] = t
    return out


def local_records(frames_dev, width, height, min_pixels=20, max_batch=16, lib=None):
    """Label this rank's frames and return their CC records + crops as host arrays (frame numbers local)."""
    n = int(frames_dev.shape[0])
    fs = device.FrameStream(width, height, max(n, 1), 2.0, 2.0, 1, min_pixels, max_batch=max_batch, lib=lib)
    try:
        if n:
            fs.push_records(frames_dev)
        r = fs.read(with_crops=True)
        return {"rec": r["rec"], "frame_off": r["frame_off"], "crop_off": r["crop_off"], "crop": r["crop"][:r["n_crop_words"]], "n": n}
    finally:
        fs.close()


def merge_records(parts):
    """Concatenate per-rank record sets in rank (= frame) order: frame numbers and crop offsets become global."""
    recs, offs, coffs, crops = [], [np.zeros(1, np.int64)], [], []
    f0 = cc0 = w0 = 0
    for p in parts:
        rec = p["rec"].copy()
        if len(rec):
            rec[:, 6] += f0
            rec[:, 7] = -1
        recs.append(rec)
        offs.append(p["frame_off"][1:] + cc0)
        coffs.append(p["crop_off"] + w0)
        crops.append(p["crop"])
        f0 += p["n"]
        cc0 += len(rec)
        w0 += len(p["crop"])
    return {"rec": np.concatenate(recs).astype(np.int32).reshape(-1, 8), "frame_off": np.concatenate(offs).astype(np.int64),
            "crop_off": np.concatenate(coffs).astype(np.int64), "crop": np.concatenate(crops).astype(np.uint32), "n_unique": 0,
            "tempo_count": 0, "active": np.zeros(0, np.int32), "active_cc": np.zeros(0, np.int32),
            "active_last": np.zeros(0, np.int32), "n_matched": 0}


def run_stream_sharded(my_frames_dev, n_frames_total, width, height, min_recall=0.85, min_precision=0.85, max_gap=85, min_pixels=20,
                       max_batch=16, lib=None):
    """Every rank passes the device frames of ITS frame_range(); rank 0 returns a matched FrameStream holding the whole
    stream (ready for device.Grouping), the other ranks return None."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    part = local_records(my_frames_dev, width, height, min_pixels, max_batch, lib)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(part, gathered, dst=0)
    if rank != 0:
        return None
    state = merge_records(gathered)
    assert len(state["frame_off"]) - 1 == n_frames_total
    fs = device.FrameStream(width, height, n_frames_total, min_recall, min_precision, max_gap, min_pixels, max_batch=max_batch,
                            max_ccs=max(len(state["rec"]), 64), max_crop_words=max(len(state["crop"]), 64), lib=lib)
    fs.import_state(state)
    fs.match(n_frames_total)
    return fs
