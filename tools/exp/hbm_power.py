"""Power of pure HBM streaming (no matrix work): a 1 GiB device-to-device copy in a loop for ~8 s, GB/s printed (bytes read + written); sample rocm-smi beside it
(tools/exp/power_probe.sh style).  python tools/exp/hbm_power.py [seconds]"""
import sys, time
import torch
sec = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
a = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < sec:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    print(f"copy: {50 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9:.0f} GB/s", flush=True)
