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


def gather_blocks(buf, dst=0, be=None, failed=False):
    """Every rank passes its packed block (flat uint8 buffer on the device; numpy with the emulated library).  Rank `dst`
    returns the list of all ranks' blocks in rank order (its own block is not copied), the others return None.  Sizes travel
    in one all_gather, payloads point to point (send / recv) -- device to device over RCCL.  A rank whose local work failed
    passes failed=True (any small buffer): it still takes part in the size exchange, and every rank raises after it."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    on_device = hasattr(buf, "data_ptr")
    nccl = dist.get_backend() == "nccl"
    t = buf if on_device else torch.from_numpy(np.ascontiguousarray(buf))
    if on_device and not nccl:
        t = t.cpu()                                   # gloo rehearsal on a GPU box: stage through the host
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([-1 if failed else t.numel()], dtype=torch.int64, device=t.device))
    bad = [r for r in range(world) if int(sizes[r].item()) < 0]
    if bad:         # every rank learns it in the same collective: nobody is left waiting in a send / recv
        raise RuntimeError("rank(s) %s failed before the gather of their block" % bad)
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
        block, err = None, None
        try:
            if n:
                if full:
                    fs.push(my_frames_dev)            # rank 0 matches its own block while the others still label theirs
                else:
                    fs.push_records(my_frames_dev)
            block = fs.pack(0, n) if not full else None
        except Exception as e:                    # a capacity error, a bad block: the other ranks must not wait for this one
            err = e
        dummy = fs.be.empty((32,), np.uint8)
        try:
            blocks = gather_blocks(dummy if (full or err is not None) else block, dst=0, be=fs.be, failed=err is not None)
        except RuntimeError:
            if err is not None:
                raise err
            raise
        if not full:
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


# ----------------------------------------------------------------------------------------------------------------------
# the pipelined form: pieces, matching on rank 0, step 03 on rank 1
# ----------------------------------------------------------------------------------------------------------------------
def piece_bounds(n, pieces):
    """[lo, hi) of `pieces` nearly equal consecutive pieces of n frames (empty ones dropped)"""
    out = []
    for c in range(pieces):
        lo, hi = n * c // pieces, n * (c + 1) // pieces
        if hi > lo:
            out.append((lo, hi))
    return out


class _PeerFailed(RuntimeError):
    """a peer sent a failure header in place of a piece"""


class _Wire:
    """point-to-point transfers of flat uint8 device buffers: device to device over RCCL ("nccl"), staged through the host under
    gloo (CPU tests, the one-GPU rehearsal).  Sends are asynchronous; `drain` waits for them (buffers stay referenced until then)."""

    def __init__(self, be):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.be = torch, dist, be
        self.nccl = dist.get_backend() == "nccl"
        self.pending = []

    def _wire_tensor(self, buf):
        t = buf if hasattr(buf, "data_ptr") else self.torch.from_numpy(np.ascontiguousarray(buf))
        return t if (self.nccl or not t.is_cuda) else t.cpu()

    def send(self, buf, dst, failed=False):
        """size header (-1: the sender failed, nothing follows) + payload"""
        dev = "cuda" if self.nccl else "cpu"
        if failed:
            self.pending.append((self.dist.isend(self.torch.tensor([-1], dtype=self.torch.int64, device=dev), dst=dst), None))
            return
        t = self._wire_tensor(buf)
        head = self.torch.tensor([t.numel()], dtype=self.torch.int64, device=dev)
        self.pending.append((self.dist.isend(head, dst=dst), head))
        self.pending.append((self.dist.isend(t, dst=dst), t))

    def recv(self, src, discard=False):
        dev = "cuda" if self.nccl else "cpu"
        head = self.torch.zeros(1, dtype=self.torch.int64, device=dev)
        self.dist.recv(head, src=src)
        n = int(head.item())
        if n < 0:
            raise _PeerFailed("rank %d reported a failure in its share of the stream" % src)
        rb = self.torch.empty(n, dtype=self.torch.uint8, device=dev)
        self.dist.recv(rb, src=src)
        if discard:
            return None
        if self.nccl:
            return rb
        if self.be.device:
            return rb.cuda()
        out = self.be.empty((n,), np.uint8)
        out[:] = rb.numpy()
        return out

    def drain(self):
        for w, _keep in self.pending:
            w.wait()
        self.pending = []


class ShardedStream:
    """ONE stream of n_frames per step over the ranks of the process group, pipelined (SURVEY.md 8(e) + what measuring it showed:
    the sequential half is not small -- at 10,000 1080p frames the matching replay is ~140 ms and step 03 ~105 ms of a ~230 ms step):
      * every rank runs the per-frame half (threshold -> label -> records / crops) of its contiguous frame range in `pieces` pieces and
        sends each piece to rank 0 as soon as it is packed (lm_stream_pack), asynchronously;
      * rank 0 matches its own pieces as it produces them, then appends and matches the other ranks' pieces in frame order
        (first-match-wins against first-seen masks is sequential, cc_stability_estimator.py:90-123) while they still label later ones;
      * the matched stream (records + crops + assignment) goes to `group_rank` (rank 1 when there is one), which runs step 03 and the
        reconstruction while the other ranks are already in the next step.
    step() returns the matched whole-stream FrameStream on group_rank and None elsewhere.  A rank that fails sends a failure header
    in place of the piece it could not produce, so rank 0 raises instead of waiting for ever."""

    def __init__(self, width, height, n_frames, batch, lib=None, pieces=4, min_recall=0.85, min_precision=0.85, max_gap=85, min_pixels=20,
                 max_ccs_per_frame=4096, max_words_per_frame=None, group_on_second_rank=True):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self.W, self.H, self.F, self.B, self.pieces = width, height, n_frames, batch, max(1, pieces)
        self.lo, self.hi = frame_range(n_frames, self.rank, self.world)
        self.group_rank = 1 if (self.world > 1 and group_on_second_rank) else 0
        words = max_words_per_frame or max(1 << 17, (width * height) // 16)

        def make(frames):
            frames = max(frames, 1)
            return device.FrameStream(width, height, frames, min_recall, min_precision, max_gap, min_pixels, max_batch=batch,
                                      max_ccs=frames * max_ccs_per_frame, max_crop_words=frames * words, lib=lib)
        self.fs = make(n_frames if self.rank == 0 else self.hi - self.lo)           # rank 0: the whole stream (it matches it)
        self.gs = make(n_frames) if (self.rank == self.group_rank and self.rank != 0) else None
        self.lib, self.be = self.fs.lib, self.fs.be
        self.wire = _Wire(self.be) if self.world > 1 else None
        self.failed = False

    def close(self):
        for s in (self.fs, self.gs):
            if s is not None:
                s.close()
        self.fs = self.gs = None

    def step(self, logits_of, labels=None, stream_wide=None, stream_match=None, schedule=1):
        """logits_of(lo, hi) -> device fp32 [hi - lo, H, W]: the logits of frames lo..hi of THIS rank's range (0-based in the range).
        labels: optional device int32 [batch, H, W] receiving the label image of every batch in turn.
        stream_wide becomes torch's current stream for the duration of the step: pack / append / match / the transfers all order
        against the backend's current stream, which must be the one the labelling and record kernels were launched on."""
        ws = self.be.stream() if stream_wide is None else stream_wide
        ms = ws if stream_match is None else stream_match
        if self.failed:
            raise RuntimeError("this ShardedStream failed in an earlier step")
        if self.be.device and stream_wide is not None:
            t = self.be.torch
            with t.cuda.stream(t.cuda.ExternalStream(ws)):
                return self._step(logits_of, labels, ws, ms, schedule)
        return self._step(logits_of, labels, ws, ms, schedule)

    def _step(self, logits_of, labels, ws, ms, schedule):
        from . import _lib
        fs, lib = self.fs, self.lib
        if self.wire:
            self.wire.drain()                   # the previous step's sends
        fs.reset()
        mine = piece_bounds(self.hi - self.lo, self.pieces)
        if self.rank != 0:
            try:
                for a, b in mine:
                    lg = logits_of(a, b)
                    lib.check(lib.lm_stream_run_logits(fs.handle, _lib.ptr(lg), b - a, self.B, None, _lib.ptr(labels), 128, 0, schedule, ws, ws))
                    self.wire.send(fs.pack(a, b - a), 0)
            except Exception:
                # ONE failure header in place of the piece that could not be produced: rank 0 receives the pieces sent so far, then the
                # header, and raises there; waiting for exactly these sends keeps the connection up until it has seen them
                self.wire.send(None, 0, failed=True)
                self.wire.drain()
                self.failed = True
                raise
            if self.rank == self.group_rank:
                try:
                    self.gs.reset()
                    self.gs.append_packed(self.wire.recv(0))        # raises when rank 0 reports a failure (its own, or another rank's)
                    self.gs.import_assign(self.wire.recv(0))
                except Exception:
                    self.failed = True
                    raise
                return self.gs
            return None
        # ---- rank 0: its own pieces, then every other rank's in frame order.  Whatever fails here -- its own share, a transfer, a
        # failure header from a peer -- the ranks that did NOT fail have sends in flight towards rank 0 and group_rank waits for the
        # matched stream: rank 0 keeps receiving (and discarding) what is still expected, tells group_rank, and only then raises.
        progress = {"rank": 1, "piece": 0, "dead": set()}
        try:
            for a, b in mine:
                lg = logits_of(a, b)
                lib.check(lib.lm_stream_run_logits(fs.handle, _lib.ptr(lg), b - a, self.B, None, _lib.ptr(labels), 128, 1, schedule, ws, ms))
            if ms != ws:
                self._join(ms, ws)
            for r in range(1, self.world):
                rlo, rhi = frame_range(self.F, r, self.world)
                for i, (a, b) in enumerate(piece_bounds(rhi - rlo, self.pieces)):
                    progress["rank"], progress["piece"] = r, i
                    try:
                        blk = self.wire.recv(r)
                    except _PeerFailed:
                        progress["dead"].add(r)
                        raise
                    progress["piece"] = i + 1
                    fs.append_packed(blk)
                    fs.match(b - a)
            progress["rank"] = self.world
            if self.group_rank != 0:
                self.wire.send(fs.pack(0, self.F), self.group_rank)
                self.wire.send(fs.export_assign(), self.group_rank)
                return None
            return fs
        except Exception:
            self.failed = True
            self._abandon_step(progress)
            raise

    def _abandon_step(self, progress):
        """rank 0 after a failure: receive and discard the pieces still expected (a peer that sent a failure header sends nothing
        after it), report the failure to group_rank unless it is the rank that failed, wait for that send"""
        for r in range(progress["rank"], self.world):
            if r in progress["dead"]:
                continue
            rlo, rhi = frame_range(self.F, r, self.world)
            n = len(piece_bounds(rhi - rlo, self.pieces))
            for _ in range(progress["piece"] if r == progress["rank"] else 0, n):
                try:
                    self.wire.recv(r, discard=True)
                except _PeerFailed:
                    progress["dead"].add(r)
                    break
                except Exception:
                    break               # the transport itself is gone: nothing more can be received from this rank
        if self.group_rank != 0 and self.group_rank not in progress["dead"]:
            try:
                self.wire.send(None, self.group_rank, failed=True)
                self.wire.drain()
            except Exception:
                pass

    def _join(self, side, main):
        """the work queued on the matching stream precedes what follows on the main one"""
        if self.be.device:
            t = self.be.torch
            ev = t.cuda.Event()
            ev.record(t.cuda.ExternalStream(side))
            t.cuda.ExternalStream(main).wait_event(ev)

    def finish(self):
        if self.wire and not self.failed:
            self.wire.drain()
