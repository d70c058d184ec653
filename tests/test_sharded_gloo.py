"""CPU, world_size 2, gloo: frame-range sharding of one stream gives exactly the single-process result, and the FCN weight
broadcast delivers identical tensors (the N > 1 path; on the GPU box the same code runs over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lecturemath_amd import _lib, device, sharded, synth
    import lm_checks
    from oracle import cc as occ
    from oracle import fcn as ofcn
    lib = _lib.load(emu_path)
    frames = np.stack(list(synth.binary_stream(21, 96, 160, seed=8, glyphs_per_add=4, erase_every=8, jitter_p=0.4, occluder=True,
                                               max_ext=16)))
    f0, f1 = sharded.frame_range(len(frames), rank, world)
    fs = sharded.run_stream_sharded(frames[f0:f1], len(frames), 160, 96, max_gap=5, max_batch=4, lib=lib)
    sd = ofcn.random_state_dict((8,) * 18, pixel_kernel=3, seed=3) if rank == 0 else None
    got = sharded.broadcast_state_dict(sd, src=0)
    ref = ofcn.random_state_dict((8,) * 18, pixel_kernel=3, seed=3)
    assert all(torch.equal(got[k], ref[k]) for k in ref) and len(got) == len(ref)
    # the pipelined form: pieces sent as they are packed, matching on rank 0, the matched stream handed to rank 1 for step 03; two steps
    logits = synth.logits_from_binary(frames, seed=2)
    sh = sharded.ShardedStream(160, 96, len(frames), 4, lib=lib, pieces=3, max_gap=5)
    assert sh.group_rank == 1
    for step in range(2):
        gs = sh.step(lambda a, b: np.ascontiguousarray(logits[f0 + a:f0 + b]))
        assert (gs is not None) == (rank == 1)
        if rank == 1:
            single = device.FrameStream(160, 96, len(frames), 0.85, 0.85, 5, 20, max_batch=4, lib=lib)
            single.push(frames)
            a, b = gs.read(), single.read()
            for key in ("rec", "frame_off", "crop_off"):
                assert (a[key] == b[key]).all(), key
            assert (a["crop"][:a["n_crop_words"]] == b["crop"][:b["n_crop_words"]]).all() and a["tempo_count"] == b["tempo_count"] and a["n_unique"] == b["n_unique"]
            ga, gb = device.Grouping(gs, max_gap=5).result(), device.Grouping(single, max_gap=5).result()
            assert ga["cc_groups"] == gb["cc_groups"] and ga["group_ages"] == gb["group_ages"]
            assert all((x == y).all() for x, y in zip(ga["clean_binary"], gb["clean_binary"]))
            single.close()
            open(os.path.join(out_dir, "ok_pipelined_%d" % step), "w").write("ok")
    sh.finish()
    sh.close()
    if rank == 0:
        o = occ.Stability(160, 96, 0.85, 0.85, 5)
        for f in frames:
            o.add_frame(f)
        lm_checks.state_equal_oracle(fs.result(), o.result())
        # grouping on the gathered stream equals grouping on a single-process stream
        single = device.FrameStream(160, 96, len(frames), 0.85, 0.85, 5, 20, max_batch=4, lib=lib)
        single.push(frames)
        a = device.Grouping(fs, max_gap=5).result()
        b = device.Grouping(single, max_gap=5).result()
        assert a["cc_groups"] == b["cc_groups"] and a["group_ages"] == b["group_ages"]
        assert all((x == y).all() for x, y in zip(a["clean_binary"], b["clean_binary"]))
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert fs is None
    dist.barrier()
    dist.destroy_process_group()


def test_frame_range():
    from lecturemath_amd import sharded
    assert [sharded.frame_range(10000, r, 8) for r in (0, 7)] == [(0, 1250), (8750, 10000)]
    assert [sharded.frame_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sharded.frame_range(2, 3, 4) == (2, 2)


def test_sharded_stream_two_ranks(emu_lib, oracle_built, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), emu_lib.path, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists() and (tmp_path / "ok_pipelined_0").exists() and (tmp_path / "ok_pipelined_1").exists()


def _failing_worker(rank, world, port, emu_path, out_dir):
    """rank 1 fails in its share of the work: nobody may be left waiting (the gather used to hang rank 0)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lecturemath_amd import _lib, sharded, synth
    lib = _lib.load(emu_path)
    # (1) the one-transfer gather: the failing rank still takes part in the size exchange, every rank raises
    raised = False
    try:
        sharded.gather_blocks(np.zeros(32, np.uint8), dst=0, failed=(rank == 1))
    except RuntimeError as e:
        raised = "failed before the gather" in str(e)
    assert raised
    # (2) the pipelined form: rank 1's producer raises on its second piece; rank 0 sees the failure header instead of a payload
    frames = np.stack(list(synth.binary_stream(12, 96, 160, seed=8, glyphs_per_add=4, max_ext=16)))
    logits = synth.logits_from_binary(frames, seed=2)
    f0, f1 = sharded.frame_range(len(frames), rank, world)
    sh = sharded.ShardedStream(160, 96, len(frames), 4, lib=lib, pieces=3, max_gap=5)
    calls = [0]

    def produce(a, b):
        calls[0] += 1
        if rank == 1 and calls[0] == 2:
            raise ValueError("frame source broke")
        return np.ascontiguousarray(logits[f0 + a:f0 + b])
    try:
        sh.step(produce)
        outcome = "returned"
    except ValueError:
        outcome = "producer error"
    except RuntimeError as e:
        outcome = "peer failure" if "reported a failure" in str(e) else "other: %s" % e
    assert outcome == ("peer failure" if rank == 0 else "producer error"), outcome
    sh.finish()
    open(os.path.join(out_dir, "failed_ok_%d" % rank), "w").write(outcome)
    dist.destroy_process_group()


def test_sharded_failure_does_not_hang(emu_lib, tmp_path):
    mp.spawn(_failing_worker, args=(2, _free_port(), emu_lib.path, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "failed_ok_0").exists() and (tmp_path / "failed_ok_1").exists()


def _failing_worker3(rank, world, port, emu_path, out_dir, who):
    """Three ranks, group rank 1, rank `who` fails in its own share.  who = 2: rank 0 meets the failure header while it receives, has to
    tell rank 1 (blocked waiting for the matched stream).  who = 0: rank 0 fails before it has received anything: it still has to take
    rank 1's and rank 2's pieces off the wire (their sends would never complete) and tell rank 1.  (both used to hang.)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from lecturemath_amd import _lib, sharded, synth
    lib = _lib.load(emu_path)
    frames = np.stack(list(synth.binary_stream(15, 96, 160, seed=8, glyphs_per_add=4, max_ext=16)))
    logits = synth.logits_from_binary(frames, seed=2)
    f0, f1 = sharded.frame_range(len(frames), rank, world)
    sh = sharded.ShardedStream(160, 96, len(frames), 4, lib=lib, pieces=3, max_gap=5)
    assert sh.group_rank == 1
    calls = [0]

    def produce(a, b):
        calls[0] += 1
        if rank == who and calls[0] == 2:
            raise ValueError("frame source broke")
        return np.ascontiguousarray(logits[f0 + a:f0 + b])
    try:
        sh.step(produce)
        outcome = "returned"
    except ValueError:
        outcome = "producer error"
    except RuntimeError as e:
        outcome = "peer failure" if "reported a failure" in str(e) else "other: %s" % e
    if who == 2:
        expect = {0: "peer failure", 1: "peer failure", 2: "producer error"}[rank]
    else:
        expect = {0: "producer error", 1: "peer failure", 2: "returned"}[rank]
    assert outcome == expect, (rank, outcome)
    assert sh.failed == (outcome != "returned")
    if sh.failed:           # a failed stream refuses another step instead of mixing the old step's pieces into it
        with pytest.raises(RuntimeError, match="failed in an earlier step"):
            sh.step(produce)
    sh.finish()
    open(os.path.join(out_dir, "failed3_%d_%d" % (who, rank)), "w").write(outcome)
    dist.destroy_process_group()


@pytest.mark.parametrize("who", [2, 0])
def test_sharded_failure_three_ranks(emu_lib, tmp_path, who):
    mp.spawn(_failing_worker3, args=(3, _free_port(), emu_lib.path, str(tmp_path), who), nprocs=3, join=True)
    assert all((tmp_path / ("failed3_%d_%d" % (who, r))).exists() for r in range(3))
