"""Per-step cycle stamps of producer and consumer waves of pw_split_k (library built with -DTTK_PROFILE)."""
import ctypes, os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
H.LIB_PATH = os.path.join(REPO, "neuralnet-tracker-traincode_amd", "libttk_hip_prof.so")
L, p = H.lib(), H.ptr
M, ci, co = 41472, 512, 512
dev = "cuda"
ydw, out = torch.randn(M, ci, device=dev), torch.empty(M, co, device=dev)
w = torch.randn(co, ci, device=dev) * 0.05
bn = torch.rand(8, ci, device=dev) + 0.5
part = torch.empty(L.partial_rows_gemm(M) * 2 * co, device=dev)
for _ in range(3):
    L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn), p(w), p(out), p(part), M, ci, co)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
L.cdll.ttk_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.cdll.ttk_debug_read(buf, 4096)
a = np.array(buf[:], dtype=np.int64)
cons = a[:32].reshape(16, 2)       # per step: before barrier, after barrier
prod = a[1024:1024 + 8 * 16].reshape(16, 8)[:, :5]
t0 = min(cons[0, 0], prod[0, 0])
print("step | consumer: compute  barrier-wait | producer: load_b+store_a  load_a  store_b  barrier-wait")
for i in range(16):
    c_comp = cons[i, 0] - (cons[i - 1, 1] if i else prod[0, 0])
    c_wait = cons[i, 1] - cons[i, 0]
    pa, pb, pl, pw = prod[i, 1] - prod[i, 0], prod[i, 2] - prod[i, 1], prod[i, 3] - prod[i, 2], prod[i, 4] - prod[i, 3]
    wA = 0
    print(f"{i:3d} | {c_comp:8d} {c_wait:8d} | waitA {wA:6d} {pa:8d} {pb:8d} {pl:8d} {pw:8d}   step total {cons[i,1] - (cons[i-1,1] if i else cons[0,1]):8d}")
