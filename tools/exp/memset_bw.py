import torch
n = 8192 * 786258
x = torch.empty(n, dtype=torch.uint8, device='cuda'); y = torch.empty(n // 4, dtype=torch.uint8, device='cuda')
def ev(fn, reps=6):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = ev(lambda: x.zero_()); print('zero_ 6.44 GB: %.3f ms  %.2f TB/s' % (ms, n / ms / 1e9))
ms = ev(lambda: x.fill_(1)); print('fill_ 6.44 GB: %.3f ms  %.2f TB/s' % (ms, n / ms / 1e9))
x32 = x.view(torch.int32)
ms = ev(lambda: x32.fill_(7)); print('fill_ int32  : %.3f ms  %.2f TB/s' % (ms, n / ms / 1e9))
