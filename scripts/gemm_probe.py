"""Development probe: the two fp32 GEMMs of the box head (hipBLASLt through torch) and split-K alternatives.
python scripts/gemm_probe.py"""
import torch, time
dev = torch.device("cuda:0")
torch.manual_seed(0)

def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

# GEMM 1: [49000, 384] x [384, 512] + bias
a = torch.randn(49000, 384, device=dev); w = torch.randn(512, 384, device=dev); b = torch.randn(512, device=dev)
wt = w.t().contiguous()
print("conv3d-as-GEMM addmm(w.t() view):", bench(lambda: torch.addmm(b, a, w.t())))
print("conv3d-as-GEMM addmm(contiguous wt):", bench(lambda: torch.addmm(b, a, wt)))
print("conv3d-as-GEMM F.linear:", bench(lambda: torch.nn.functional.linear(a, w, b)))
print("conv3d-as-GEMM mm only:", bench(lambda: torch.mm(a, wt)))
# GEMM 2: [1000, 25088] x [25088, 512] + bias
x = torch.randn(1000, 25088, device=dev); w6 = torch.randn(512, 25088, device=dev); b6 = torch.randn(512, device=dev)
w6t = w6.t().contiguous()
ref = torch.addmm(b6, x, w6.t())
print("fc6 addmm(w.t() view):", bench(lambda: torch.addmm(b6, x, w6.t())))
print("fc6 addmm(contiguous wt):", bench(lambda: torch.addmm(b6, x, w6t)))
for S in (2, 4, 7, 8):
    if 25088 % S: continue
    ks = 25088 // S
    w_s = w6.view(512, S, ks).permute(1, 2, 0).contiguous()          # [S, ks, 512]
    def f():
        xs = x.view(1000, S, ks).permute(1, 0, 2)                     # [S, 1000, ks] (strided view)
        return torch.bmm(xs, w_s).sum(0) + b6
    err = (f() - ref).abs().max().item() / ref.abs().max().item()
    print(f"fc6 split-K {S} (bmm + sum): {bench(f):.1f} us   rel diff {err:.2e}")
    def g():
        xs = x.view(1000, S, ks).permute(1, 0, 2)
        out = torch.baddbmm(b6.view(1, 1, 512).expand(S, 1000, 512) * 0, xs, w_s)
        return out.sum(0) + b6
