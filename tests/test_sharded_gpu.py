"""BASELINE configs[3] on ONE GPU: two fresh child processes share cuda:0 (gloo for the rendezvous, as in
`LM_BENCH_REHEARSE=1 bench.py --gpus 2`) and run exactly the code the multi-GPU bench runs with the real HIP library -- every
rank labels its contiguous block of frames, the packed records (lm_stream_pack) go to rank 0 in one transfer per rank, rank 0
appends them (lm_stream_append_packed), replays the matching and runs step 03 -- and the result equals the single-process
stream bit for bit (records, crops, assignments, groups, ages, reconstructed frames).  Over RCCL the same calls move device
buffers; that path needs >= 2 GPUs and is exercised by the driver's scaling run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lecturemath_amd import _lib, device, digests, sharded, synth
    lib = _lib.load()
    assert lib.is_device_build
    h, w, n = 540, 960, 150
    frames = np.stack(list(synth.binary_stream(n, h, w, seed=31, glyphs_per_add=20, erase_every=40, jitter_p=0.1)))
    f0, f1 = sharded.frame_range(n, rank, world)
    mine = torch.from_numpy(frames[f0:f1]).cuda()
    fs = sharded.run_stream_sharded(mine, n, w, h, max_gap=85, max_batch=32, lib=lib)
    # FCN weights: one contiguous broadcast
    from oracle import fcn as ofcn
    sd = ofcn.random_state_dict((8,) * 18, pixel_kernel=3, seed=3) if rank == 0 else None
    got = sharded.broadcast_state_dict(sd, src=0, device_name="cuda")
    ref = ofcn.random_state_dict((8,) * 18, pixel_kernel=3, seed=3)
    assert len(got) == len(ref) and all(torch.equal(got[k].cpu(), ref[k]) for k in ref)
    # the pipelined form (bench.py --gpus N): pieces to rank 0 as they are packed, matching on rank 0, the MATCHED stream handed to rank 1
    # (lm_stream_pack + lm_stream_export_assign -> lm_stream_append_packed + lm_stream_import_assign), step 03 there; two steps
    logits = torch.from_numpy(synth.logits_from_binary(frames[f0:f1], seed=4)).cuda()
    sh = sharded.ShardedStream(w, h, n, 32, lib=lib, pieces=3)
    for step in range(2):
        gs = sh.step(lambda a, b: logits[a:b])
        assert (gs is not None) == (rank == 1)
        if rank == 1:
            single = device.FrameStream(w, h, n, 0.85, 0.85, 85, 20, max_batch=32, lib=lib)
            single.push(torch.from_numpy(frames).cuda())
            a, b = gs.read(), single.read()
            for key in ("rec", "frame_off", "crop_off"):
                assert (a[key] == b[key]).all(), key
            assert (a["crop"][:a["n_crop_words"]] == b["crop"][:b["n_crop_words"]]).all() and a["tempo_count"] == b["tempo_count"]
            ga, gb = device.Grouping(gs), device.Grouping(single)
            assert digests.from_device(gs, ga) == digests.from_device(single, gb)
            assert bool((ga.render(0, n) == gb.render(0, n)).all())
            ga.close(); gb.close(); single.close()
            open(os.path.join(out_dir, "ok_pipelined_%d" % step), "w").write("ok")
    sh.finish()
    sh.close()
    if rank == 0:
        single = device.FrameStream(w, h, n, 0.85, 0.85, 85, 20, max_batch=32, lib=lib)
        single.push(torch.from_numpy(frames).cuda())
        a, b = fs.read(), single.read()
        for key in ("rec", "frame_off", "crop_off", "active"):
            assert (a[key] == b[key]).all(), key
        assert (a["crop"][:a["n_crop_words"]] == b["crop"][:b["n_crop_words"]]).all() and a["tempo_count"] == b["tempo_count"]
        ga, gb = device.Grouping(fs), device.Grouping(single)
        assert digests.from_device(fs, ga) == digests.from_device(single, gb)
        assert bool((ga.render(0, n) == gb.render(0, n)).all())
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert fs is None
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stream_two_processes_on_one_gpu(hip_lib, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists() and (tmp_path / "ok_pipelined_0").exists() and (tmp_path / "ok_pipelined_1").exists()
