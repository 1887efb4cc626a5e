"""Write-only / copy / read-only streams over 1 GiB with torch on the GPU box (round 6: is a store-dominated kernel at 2.2 TB/s
at the chip's write limit?  No: fill 6.9 TB/s, copy 5.5 TB/s on MI355X)."""
import torch, time
x = torch.empty(256*1024*1024, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
for name, fn, nbytes in (("fill (write only)", lambda: x.fill_(1.0), x.numel()*4), ("copy (read + write)", lambda: y.copy_(x), 2*x.numel()*4),
                         ("sum (read only)", lambda: x.sum(), x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-22s %.3f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
