import torch
x = torch.empty(512*65*65*32, device="cuda")
for n in (x.numel(), x.numel()*2):
    y = torch.empty(n, device="cuda")
    for _ in range(3): y.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("fill", n * 4 / 1e6, "MB", round(us, 1), "us", round(n * 4 / us / 1e6, 2), "TB/s")
