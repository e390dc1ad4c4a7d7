import torch, time
dev = "cuda"
F, H, W = 64, 480, 640
def timed(fn, reps=50, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
x = torch.rand((F, H, W), device=dev)
y = torch.empty((F, H, W, 2), device=dev)
xe = x.unsqueeze(-1).expand(-1, -1, -1, 2)
t = timed(lambda: y.copy_(xe))
print(f"read 4 B + write 8 B per px ({x.numel()*12/1e6:.0f} MB): {t:.1f} us = {x.numel()*12/t/1e3:.0f} GB/s")
z = torch.empty_like(y)
t = timed(lambda: z.fill_(1.0))
print(f"write-only 157 MB: {t:.1f} us = {z.numel()*4/t/1e3:.0f} GB/s")
big = torch.rand((256, 1024, 1024), device=dev); big2 = torch.empty_like(big)
t = timed(lambda: big2.copy_(big), reps=10, warm=5)
print(f"copy 1 GB -> 1 GB: {t:.1f} us = {big.numel()*8/t/1e3:.0f} GB/s")
t = timed(lambda: big.sum(), reps=10, warm=5)
print(f"read-only 1 GB (sum): {t:.1f} us = {big.numel()*4/t/1e3:.0f} GB/s")
# cold variant of the mix: rotate over 4 distinct buffer sets (1 GB total) so that nothing stays in the Infinity Cache
xs = [torch.rand((F, H, W), device=dev) for _ in range(4)]; ys = [torch.empty((F, H, W, 2), device=dev) for _ in range(4)]
i = [0]
def rot():
    k = i[0] % 4; i[0] += 1
    ys[k].copy_(xs[k].unsqueeze(-1).expand(-1, -1, -1, 2))
t = timed(rot, reps=48, warm=16)
print(f"mix, 4 rotating buffer sets: {t:.1f} us = {x.numel()*12/t/1e3:.0f} GB/s")
