"""CPU, world_size 2, gloo: the N>1 bookkeeping of the sharded evaluation (shards, gather order, metric reduce)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zutis_amd import distributed as zd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_images, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        full = torch.randn((n_images, 5, 6, 7), generator=g)            # "low-res logits" of the whole eval set
        labels = torch.randint(0, 5, (n_images, 12, 9), generator=g)
        preds = torch.randint(0, 5, (n_images, 12, 9), generator=g)
        lo, hi = zd.shard_range(n_images, rank, world)
        ragged = zd.all_gather_ragged(full[lo:hi], n_images)
        assert torch.equal(ragged, full)
        b = n_images // world                                            # equal shards -> plain gather, async as in bench.py
        out, work = zd.all_gather_logits(full[rank * b:(rank + 1) * b], async_op=True)
        work.wait()
        assert torch.equal(out, full[: world * b])
        hist = torch.bincount((5 * labels[lo:hi] + preds[lo:hi]).reshape(-1), minlength=25).reshape(5, 5)
        zd.all_reduce_confusion(hist)
        ref = torch.bincount((5 * labels + preds).reshape(-1), minlength=25).reshape(5, 5)
        assert torch.equal(hist, ref)
        # sharded retrieval (config 5): local top-k per shard, all-gather of the candidates, identical merge on every rank.
        # The local top-k is a GPU kernel in the product; here the oracle stands in for it so that the collective + merge
        # bookkeeping (offsets, padding of short shards, tie order) runs on CPU.
        from oracle import zutis_ref as O
        from zutis_amd import retrieval
        text = torch.randn((4, 16), generator=g)
        images = torch.randn((n_images + 4, 16), generator=g)
        images[5] = images[2]                                            # exact score tie across shards: smaller index first
        def cpu_topk(t, im, k, index_offset=0):
            i, v = O.retrieve_topk(t.numpy(), im.numpy(), k)
            return torch.from_numpy(i) + index_offset, torch.from_numpy(v)

        def cpu_merge(ci, cv, k):                                        # the product's merge is the zh_topk_rows kernel
            i, v = O.merge_topk(ci.numpy(), cv.numpy(), k)
            return torch.from_numpy(i), torch.from_numpy(v)
        retrieval.retrieve_topk, retrieval.merge_topk = cpu_topk, cpu_merge
        lo2, hi2 = zd.shard_range(images.shape[0], rank, world)
        k = 6                                                            # > the smaller shard: exercises the -1 / -inf padding
        idx, val = retrieval.retrieve_topk_sharded(text, images[lo2:hi2], lo2, k)
        ri, rv = cpu_topk(text, images, k)
        assert torch.equal(idx, ri) and torch.allclose(val, rv)
        # k larger than the GLOBAL image count is clamped exactly as the unsharded path clamps it (no -1 rows leak out)
        idx, val = retrieval.retrieve_topk_sharded(text, images[lo2:hi2], lo2, images.shape[0] + 5)
        ri, rv = cpu_topk(text, images, images.shape[0])
        assert idx.shape == ri.shape and torch.equal(idx, ri) and int(idx.min()) >= 0
        # evaluation pipeline (bench.py's scheduling): 3 lanes, 7 steps (ragged final group of 1), async gathers retired one
        # round later; every gathered buffer must hold both ranks' payloads of ITS step, in rank order
        seen = []

        def launch(grp, ids):
            for ln, i in zip(grp, ids):
                ln.payload.copy_(torch.full((2, 3), float(100 * i), dtype=torch.float32) + rank)

        def consume(ln, step):
            exp = torch.cat([torch.full((2, 3), float(100 * step + r)) for r in range(world)])
            assert torch.equal(ln.gathered, exp), (step, ln.gathered)
            seen.append(step)
        lanes = [zd.Lane(torch.zeros((2, 3)), gathered=torch.empty((2 * world, 3))) for _ in range(3)]
        pipe = zd.StepPipeline(lanes, launch, gather=True, consume=consume)
        pipe.run(3)                                                      # warm-up group
        pipe.run(7)
        pipe.drain()
        assert sorted(seen) == list(range(10)) and pipe.next_step == 10 and pipe.retired == 10
        # bench.py's proof that the timed gathers saw N ranks (benchlib/headline.py runs exactly this after pipe.drain()): every rank's
        # payload checksum, all-gathered on its own, against the checksums of the slices of lane 0's gathered buffer
        v = zd.verify_gather(lanes[0])
        assert v["verified"] and v["slices_distinct"] and v["ranks_in_gather"] == world and v["bytes_per_rank"] == 24 and v["step"] == lanes[0].step, v
        keep = lanes[0].gathered.clone()
        lanes[0].gathered[2 * (1 - rank)] += 1.0                        # the OTHER rank's slice is damaged (on both ranks): caught
        v = zd.verify_gather(lanes[0])
        assert not v["verified"] and v["mismatching_slices_on_this_rank"] == [1 - rank], v
        lanes[0].gathered.copy_(keep)
        if rank == 1:                                                    # damage on ONE rank only: the verdict is min-reduced, rank 0 sees it too
            lanes[0].gathered[0, 0] = 7.0
        v = zd.verify_gather(lanes[0])
        assert not v["verified"] and v["mismatching_slices_on_this_rank"] == ([0] if rank == 1 else []), v
        lanes[0].gathered.copy_(torch.cat([lanes[0].payload] * world))   # a "gather" that only ever saw this rank's buffer
        v = zd.verify_gather(lanes[0])
        assert not v["verified"] and v["mismatching_slices_on_this_rank"] == [1 - rank], v
        lanes[0].gathered.copy_(keep)
        same = zd.Lane(torch.ones((2, 3)), gathered=torch.ones((2 * world, 3)))       # identical data on every rank: verified, but flagged
        same.step = 0
        v = zd.verify_gather(same)
        assert v["verified"] and not v["slices_distinct"], v
        nog = zd.StepPipeline([zd.Lane(torch.zeros(1)) for _ in range(2)], lambda g_, ids: seen.extend(-i - 1 for i in ids), gather=False)
        nog.run(5); nog.drain()
        assert [s for s in seen if s < 0] == [-1, -2, -3, -4, -5]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [zd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_gloo_gather_and_reduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "ok"), (1, "ok")], res
