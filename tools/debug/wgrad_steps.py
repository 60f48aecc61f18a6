import ctypes, os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
H.LIB_PATH = os.path.join(REPO, "neuralnet-tracker-traincode_amd", "libttk_hip_prof.so")
L, p = H.lib(), H.ptr
M, ci, co = 41472, 512, 512
dev = "cuda"
ydw, y, g = torch.randn(M, ci, device=dev), torch.randn(M, co, device=dev), torch.randn(M, co, device=dev)
bn_dw, bn_pw = torch.rand(8, ci, device=dev) + 0.5, torch.rand(8, co, device=dev) + 0.5
dW = torch.zeros(co, ci, device=dev)
for _ in range(3):
    L.call("ttk_pwconv1x1_bwd_weight", p(g), p(y), p(bn_pw), p(ydw), p(bn_dw), p(dW), M, ci, co)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
L.cdll.ttk_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.cdll.ttk_debug_read(buf, 4096)
a = np.array(buf[:], dtype=np.int64)
prod = a[1024:1024 + 8 * 24].reshape(24, 8)[:, :6]
print("step | store_a(+wait) load_a store_b(+wait) load_b barrier | total")
for i in range(1, 20):
    d = np.diff(prod[i])
    print(f"{i:3d} | {d[0]:7d} {d[1]:7d} {d[2]:7d} {d[3]:7d} {d[4]:7d} | {prod[i,5]-prod[i-1,5]:7d}")
