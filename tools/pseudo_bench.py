"""Developer tool: pseudo-label driver throughput (SelfMask + bilateral solver + resize + RLE JSON), 1 vs n streams."""
import sys, os, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zutis_amd import detgen, pseudo_masks
from zutis_amd.engine import SelfMaskEngine
dev = torch.device("cuda:0")
eng = SelfMaskEngine({k: torch.from_numpy(v).to(dev) for k, v in detgen.selfmask_state_dict().items()})
N, H, W = 24, 384, 512
imgs = [torch.from_numpy(detgen.selfmask_like_rgb(H, W, seed=i)).to(dev) if False else torch.from_numpy(detgen.images(1, H, W, seed=i))[0].to(dev) for i in range(N)]
sizes = [(480, 640)] * N
with tempfile.TemporaryDirectory() as d:
    for S in (1, 2, 4, 8):
        paths = [os.path.join(d, f"s{S}", f"{i}.json") for i in range(N)]
        pseudo_masks.generate_pseudo_masks(eng, imgs[:4], sizes[:4], paths[:4], n_streams=S)      # warm-up
        torch.cuda.synchronize(); t = time.perf_counter()
        pseudo_masks.generate_pseudo_masks(eng, imgs, sizes, paths, n_streams=S)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"streams={S}: {N/dt:.1f} images/s ({dt/N*1e3:.2f} ms/image)")
    a = open(os.path.join(d, "s1", "5.json")).read(); b = open(os.path.join(d, "s4", "5.json")).read()
    print("same output 1 vs 4 streams:", a == b)

# device-side pipeline only (random-init SelfMask gives noise masks whose RLE is pathologically long: the numbers above are
# dominated by the host decode of ~150 k runs; real masks have a few hundred)
for S in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    engines = [eng] + [eng.fork() for _ in range(S - 1)]
    def run():
        outs = []
        for i, img in enumerate(imgs):
            k = i % S
            with torch.cuda.stream(streams[k]):
                outs.append(pseudo_masks._device_mask(engines[k], img, (480, 640), True))
        return outs
    run(); torch.cuda.synchronize(); t = time.perf_counter()
    run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"device only, streams={S}: {N/dt:.1f} images/s ({dt/N*1e3:.2f} ms/image)")
