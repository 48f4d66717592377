"""HBM ceilings on this box for the access patterns the kernels use (torch fill / copy / add as the yardstick)."""
import torch, time
n = 1610612736 // 2      # bf16 elements of a [64*128*128, 768] activation
a = torch.empty(n, dtype=torch.bfloat16, device='cuda'); b = torch.empty_like(a); c = torch.empty_like(a)
def t(fn, byt, name, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / it
    print(f'{name:28s} {dt * 1e3:8.3f} ms  {byt / dt / 1e12:6.2f} TB/s')
t(lambda: a.zero_(), 2 * n, 'fill (write only)')
t(lambda: b.copy_(a), 4 * n, 'copy (1R + 1W)')
t(lambda: torch.add(a, b, out=c), 6 * n, 'add (2R + 1W)')
t(lambda: a.sum(), 2 * n, 'sum (read only)')
