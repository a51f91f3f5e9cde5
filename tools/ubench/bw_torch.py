import torch, time
x = torch.empty(256*1024*1024, dtype=torch.float32, device='cuda')  # 1 GiB
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.fill_(1.0)); print('fill 1 GiB: %.3f ms = %.2f TB/s write' % (ms, 1.0737 / ms))
ms = t(lambda: x.sum()); print('sum 1 GiB: %.3f ms = %.2f TB/s read' % (ms, 1.0737 / ms))
ms = t(lambda: y.copy_(x)); print('copy 1 GiB: %.3f ms = %.2f TB/s read+write' % (ms, 2 * 1.0737 / ms))
ms = t(lambda: torch.add(x, 1.0, out=y)); print('add 1 GiB: %.3f ms = %.2f TB/s read+write' % (ms, 2 * 1.0737 / ms))
