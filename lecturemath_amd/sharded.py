"""Frame-range sharding of ONE stream across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests and the one-GPU rehearsal).

The per-frame stages (threshold, labelling, statistics, CC records + crops; labeler.py:116-191) are independent per frame and
run on the rank that owns the frame range; temporal matching (cc_stability_estimator.py:71-145) carries state from frame to
frame and first-match-wins against FIRST-SEEN masks, so it is replayed sequentially on rank 0 over the gathered records --
KBs per frame.  Data-path collectives: one all_gather of block sizes and one point-to-point transfer per rank
of its packed block (lm_stream_pack: records + crops as one flat device buffer, no pickling); plus the one-off broadcast of
the FCN weights as ONE contiguous buffer.
"""
import numpy as np

from . import device


def frame_range(n_frames, rank, world):
    """Contiguous block of ceil(F/R) frames per rank."""
    per = -(-n_frames // world)
    return min(rank * per, n_frames), min((rank + 1) * per, n_frames)


def _comm_device(device_name):
    import torch.distributed as dist
    return device_name if (device_name and dist.get_backend() == "nccl") else "cpu"


def broadcast_state_dict(sd, src=0, device_name=None):
    """Rank `src` holds the FCN state_dict (147 tensors, 63 MB fp32 at the shipped widths); every rank gets a copy.
    One broadcast of the (key, shape, dtype) table and ONE broadcast of all tensors flattened into a contiguous byte buffer
    (over RCCL the buffer lives on the device)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    meta = [[(k, tuple(sd[k].shape), str(sd[k].dtype).replace("torch.", "")) for k in sorted(sd.keys())]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)         # control plane: names and shapes only
    meta = meta[0]
    sizes = [int(np.prod(shape)) * torch.empty((), dtype=getattr(torch, dt)).element_size() for _, shape, dt in meta]
    offs = np.concatenate([[0], np.cumsum([(n + 15) & ~15 for n in sizes])]).astype(np.int64)
    cdev = _comm_device(device_name)
    flat = torch.zeros(int(offs[-1]), dtype=torch.uint8, device=cdev)
    if rank == src:
        for (k, shape, dt), o, n in zip(meta, offs[:-1], sizes):
            if n:
                flat[int(o):int(o) + n] = sd[k].detach().contiguous().reshape(-1).view(torch.uint8).to(cdev)
    dist.broadcast(flat, src=src)
    out = {}
    for (k, shape, dt), o, n in zip(meta, offs[:-1], sizes):
        t = flat[int(o):int(o) + n].clone().view(getattr(torch, dt)).reshape(shape)
        out[k] = t.to(device_name) if device_name else t
    return out


def gather_blocks(buf, dst=0, be=None):
    """Every rank passes its packed block (flat uint8 buffer on the device; numpy with the emulated library).  Rank `dst`
    returns the list of all ranks' blocks in rank order (its own block is not copied), the others return None.  Sizes travel
    in one all_gather, payloads point to point (send / recv) -- device to device over RCCL."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    on_device = hasattr(buf, "data_ptr")
    nccl = dist.get_backend() == "nccl"
    t = buf if on_device else torch.from_numpy(np.ascontiguousarray(buf))
    if on_device and not nccl:
        t = t.cpu()                                   # gloo rehearsal on a GPU box: stage through the host
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.numel()], dtype=torch.int64, device=t.device))
    if rank != dst:
        dist.send(t, dst=dst)
        return None
    out = []
    for r in range(world):
        if r == dst:
            out.append(buf)
            continue
        n = int(sizes[r].item())
        if on_device and nccl:
            rb = torch.empty(n, dtype=torch.uint8, device=t.device)
            dist.recv(rb, src=r)
        else:
            rb = torch.empty(n, dtype=torch.uint8)
            dist.recv(rb, src=r)
            if on_device:
                rb = rb.cuda()
            else:
                dev = be.empty((n,), np.uint8) if be is not None else np.empty(n, np.uint8)
                dev[:] = rb.numpy()
                rb = dev
        out.append(rb)
    return out


def run_stream_sharded(my_frames_dev, n_frames_total, width, height, min_recall=0.85, min_precision=0.85, max_gap=85, min_pixels=20,
                       max_batch=16, lib=None, max_ccs=None, max_crop_words=None):
    """Every rank passes the device frames of ITS frame_range(); rank 0 returns a matched FrameStream holding the whole
    stream (ready for device.Grouping), the other ranks return None."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    n = int(my_frames_dev.shape[0])
    full = rank == 0
    fs = device.FrameStream(width, height, n_frames_total if full else max(n, 1), min_recall, min_precision, max_gap, min_pixels,
                            max_batch=max_batch, max_ccs=max_ccs, max_crop_words=max_crop_words, lib=lib)
    try:
        if n:
            if full:
                fs.push(my_frames_dev)            # rank 0 matches its own block while the others still label theirs
            else:
                fs.push_records(my_frames_dev)
        block = fs.pack(0, n) if not full else None
        if full:
            blocks = gather_blocks(fs.be.empty((32,), np.uint8), dst=0, be=fs.be)
        else:
            gather_blocks(block, dst=0, be=fs.be)
            return None
        for r in range(1, world):
            fs.append_packed(blocks[r])
            f0, f1 = frame_range(n_frames_total, r, world)
            fs.match(f1 - f0)
        assert fs.counters()["n_frames"] == n_frames_total
        out, fs = fs, None
        return out
    finally:
        if fs is not None:
            fs.close()
