"""Per-wave cycle accounting of the full-width GEMM (csrc/pwconv_x.hip built with -DTTK_X_STAMP: tools/exp/build_variants.sh).
  TTK_LIB=tools/exp/_build/libttk_x_stamp.so python tools/exp/x_stamps.py [B]
Medians over workgroups and waves: prologue / main loop / epilogue cycles; inside the main loop the cycles spent in the waits for the activation loads,
in the waits for the weight pieces, at the barriers and in conversion + load issue; the in-kernel clock (cycles / s_memrealtime ticks x 100 MHz)."""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H  # noqa: E402
L, p = H.lib(), H.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rd = L.cdll.ttk_debug_read_x_stamps
rd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
dev = "cuda"
for name, hw, ci, co in [("dw4_1", 17, 256, 256), ("dw5_x", 9, 512, 512), ("dw6", 5, 1024, 1024)]:
    M = B * hw * hw
    ydw, y, g = torch.randn(M, ci, device=dev), torch.randn(M, co, device=dev), torch.randn(M, co, device=dev) * 1e-3
    w = torch.randn(co, ci, device=dev) * (2.0 / co) ** 0.5
    bn_dw, bn_pw = torch.rand(8, ci, device=dev) + 0.5, torch.rand(8, co, device=dev) + 0.5
    bn_dw[2], bn_pw[2], bn_pw[6] = 0.1, 0.1, 0.0
    bn_dw[7], bn_pw[7] = 0.0, 0.0
    bn_dw[7, 0], bn_pw[7, 1] = 12.0, 0.05
    out, gdw = torch.empty(M, co, device=dev), torch.empty(M, ci, device=dev)
    prep = torch.empty(L.pwconv_prepared_bytes(ci, co), dtype=torch.uint8, device=dev)
    L.pwconv_prepare_weights([w], [prep])
    part = torch.empty(max(L.partial_rows_gemm(M, ci, co), L.partial_rows_gemm(M, co, ci, True)) * 2 * max(ci, co), device=dev)
    calls = {"fwd": lambda: L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), None, p(out), p(part), None, M, ci, co, p(prep), 0),
             "dgrad": lambda: L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bn_pw), None, p(ydw), p(bn_dw), p(gdw), p(part), M, ci, co, p(prep), 0)}
    for k, fn in calls.items():
        K = ci if k == "fwd" else co
        if not L.cdll.ttk_pwconv_tile_rows(M, K, co if k == "fwd" else ci, int(k == "dgrad")):
            continue
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        junk = np.zeros(8, dtype=np.uint64)
        rd(junk.ctypes.data, junk.nbytes)  # (clears the device buffer)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
        assert rd(buf.ctypes.data, buf.nbytes) == 0
        s = buf.reshape(1024, 4, 8).astype(np.int64)
        s = s[s[:, 0, 1] > 0]
        med = lambda i: float(np.median(s[:, :, i]))
        nks = K // 32
        clock = np.median(s[:, :, :3].sum(-1) / np.maximum(s[:, :, 7], 1)) * 100.0
        print(f"== {name} {k} M={M} K={K} tile px {L.cdll.ttk_pwconv_tile_rows(M, K, co if k == 'fwd' else ci, int(k == 'dgrad'))} launch {e0.elapsed_time(e1) * 1e3:.1f} us, {len(s)} workgroups, clock {clock:.0f} MHz")
        print(f"   prologue {med(0):.0f}  loop {med(1):.0f} ({med(1) / nks:.0f} per k32 step)  epilogue {med(2):.0f} cycles")
        print(f"   per step: unit waits {med(3) / nks:.0f}  weight-piece waits {med(4) / nks:.0f}  barriers {med(5) / nks:.0f}  conversion + load issue {med(6) / nks:.0f}")
