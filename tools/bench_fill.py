"""What does this box reach for plain writes and plain copies?  (The helper-free decoders write 23.6 GB in ~8 ms.)"""
import torch
d = torch.device("cuda:0")
x = torch.empty(23_592_960_000 // 2, dtype=torch.int16, device=d)
y = torch.empty_like(x)
def t(f, n=5):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
gb = x.numel() * 2 / 1e9
ms = t(lambda: x.zero_()); print("fill   %.1f GB in %.2f ms = %.2f TB/s written" % (gb, ms, gb / ms))
ms = t(lambda: x.fill_(7)); print("fill_  %.1f GB in %.2f ms = %.2f TB/s written" % (gb, ms, gb / ms))
ms = t(lambda: y.copy_(x)); print("copy   %.1f GB in %.2f ms = %.2f TB/s read + written" % (gb, ms, 2 * gb / ms))
v = x.view(4096, -1)
ms = t(lambda: v[:, ::1].add_(1)); print("add_   %.1f GB in %.2f ms = %.2f TB/s read + written" % (gb, ms, 2 * gb / ms))
