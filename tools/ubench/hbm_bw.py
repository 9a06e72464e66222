import torch, time
x = torch.empty(512*1024*1024//8, dtype=torch.float64, device="cuda").normal_()
y = torch.empty_like(x)
for name, fn, bytes_ in (("copy (r+w)", lambda: y.copy_(x), 2*x.numel()*8), ("fill (w)", lambda: y.fill_(1.0), x.numel()*8), ("sum (r)", lambda: x.sum(), x.numel()*8), ("axpy y+=2x (2r+w)", lambda: y.add_(x, alpha=2.0), 3*x.numel()*8)):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/50
    print(f"{name:22s} {bytes_/dt/1e12:.2f} TB/s")
