"""Child process of tests/test_rccl_gpu.py: ONE rank on cuda:0 with the RCCL ("nccl") backend driving the same code path
bench.py takes for N > 1 — three lanes (engine fork + native launch plan + HIP stream each), StepPipeline with gather=True,
i.e. an asynchronous all_gather_into_tensor of the low-res class logits per step on the lane's stream.  With world_size 1 the
gathered tensor must equal the payload of the same step, bit for bit, for every step incl. the ragged last group."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
    from zutis_amd import detgen, ops, plan as zplan, distributed as zd
    from zutis_amd.engine import ZutisEngine
    cfg = detgen.TINY
    B, H, W, n = 2, 80, 112, 7
    P = {k: torch.from_numpy(v).to(dev) for k, v in detgen.zutis_state_dict(cfg).items()}
    text = torch.from_numpy(detgen.text_embeddings(n, cfg.embed_dim)).to(dev)
    eng = ZutisEngine(P, cfg.patch, cfg.dec_heads, precision="exact")
    xs = [torch.from_numpy(detgen.images(B, H, W, seed=s)).to(dev) for s in range(3)]
    ref_eng = eng.fork()
    want = [ref_eng.semantic_logits_lowres(ref_eng.forward(x)["patch_tokens"], text).clone() for x in xs]
    lanes = []
    for li in range(3):
        e = eng if li == 0 else eng.fork()
        e.forward(xs[li])
        with zplan.Recorder() as rec:
            lo = e.semantic_logits_lowres(e.forward(xs[li])["patch_tokens"], text)
            labels = torch.empty((B, H, W), dtype=torch.int64, device=dev)
            ops.upsample_argmax(lo, labels, B, n, lo.shape[2], lo.shape[3], H, W)
        lanes.append(zd.Lane(lo.view(B, n, -1), gathered=torch.empty((B, n, lo.shape[2] * lo.shape[3]), dtype=torch.float32, device=dev),
                             stream=torch.cuda.Stream(device=dev), state={"plan": rec.build(), "labels": labels, "i": li}))
    torch.cuda.synchronize()
    seen = []

    def launch(grp, ids):
        zplan.run_many([ln.state["plan"] for ln in grp], [ln.stream.cuda_stream for ln in grp])

    def consume(lane, step):
        torch.cuda.synchronize()
        assert torch.equal(lane.gathered, want[lane.state["i"]].view_as(lane.gathered)), f"gathered logits of step {step} differ"
        seen.append(step)

    pipe = zd.StepPipeline(lanes, launch, gather=True, consume=consume)
    pipe.run(7)                     # 3 + 3 + a ragged group of 1
    pipe.drain()
    torch.cuda.synchronize()
    assert sorted(seen) == list(range(7)), seen
    full = zd.all_gather_logits(want[0])
    assert torch.equal(full, want[0])
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_SMOKE_OK steps", len(seen), "backend nccl world 1", flush=True)


if __name__ == "__main__":
    main()
