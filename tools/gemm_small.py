"""Developer tool: tile/ring variants on the latency-bound (few-tile) GEMMs.  Checks each against torch fp32 too."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from zutis_amd import ops
    dev = torch.device("cuda:0")
    shapes = [("b1_qkv", 442, 2304, 768), ("b1_out", 442, 768, 768), ("b1_fc", 442, 3072, 768), ("b1_proj", 442, 768, 3072),
              ("b1_ffn1a", 1764, 2048, 768), ("b1_ffn1b", 1764, 2048, 2048), ("b1_kv", 1764, 1536, 768), ("b1_dq", 100, 768, 768),
              ("b4_qkv", 1768, 2304, 768), ("b4_out", 1768, 768, 768),
              ("d32_q", 3200, 768, 768), ("d32_ff1", 3200, 2048, 768), ("d32_ff2", 3200, 768, 2048), ("d32_ffn2", 19200, 2048, 768)]
    torch.manual_seed(0)
    for name, M, N, K in shapes:
        A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        for _ in range(5): ops.gemm(A, W, out)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): ops.gemm(A, W, out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
        err = (out.float() - A.float() @ W.float().t()).abs().max().item()
        print(f"{name:9s} {dt*1e6:7.1f} us {2*M*N*K/dt/1e12:6.1f} TF err {err:.1e}", flush=True)
else:
    for tile in ["auto", "128", "2128", "2064", "3064", "64"]:
        env = dict(os.environ)
        if tile != "auto": env["ZH_GEMM_TILE"] = tile
        print("== tile", tile, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
