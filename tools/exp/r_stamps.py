"""Per-wave cycle accounting of the row-block GEMM (csrc/pwconv_r.hip built with -DTTK_R_STAMP: tools/exp/build_variants.sh).
  TTK_LIB=tools/exp/_build/libttk_stamp.so python tools/exp/r_stamps.py [B]
Prints, per shape and direction, medians over workgroups of: prologue / main loop / epilogue cycles, cycles spent waiting in the
main loop's barriers, the auxiliary span (producers: store_a = BatchNorm form + split + ds_write; consumers: the vmcnt(0) wait for
their LDS-DMA pieces), and the in-kernel clock (cycles / s_memrealtime ticks x 100 MHz)."""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H  # noqa: E402

L, p = H.lib(), H.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda"
rd = L.cdll.ttk_debug_read_r_stamps
rd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
shapes = [("dw4_2", 9, 256, 512), ("dw5_x", 9, 512, 512), ("dw5_6", 5, 512, 1024), ("dw6", 5, 1024, 1024)]
for name, hw, ci, co in shapes:
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
    calls = {
        "fwd": (lambda: L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), None, p(out), p(part), None, M, ci, co, p(prep), 0), L.partial_rows_gemm(M, ci, co) * (co // 256)),
        "dgrad": (lambda: L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bn_pw), None, p(ydw), p(bn_dw), p(gdw), p(part), M, ci, co, p(prep), 0),
                  L.partial_rows_gemm(M, co, ci, True) * (ci // 256)),
    }
    for k, (fn, tiles) in calls.items():
        if tiles == 0:
            continue
        for _ in range(20):  # warm: clocks settle under load
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        buf = np.zeros(2048 * 12 * 8, dtype=np.uint64)
        rc = rd(buf.ctypes.data, buf.nbytes)
        assert rc == 0, rc
        s = buf.reshape(2048, 12, 8)[: min(tiles, 2048)].astype(np.int64)
        print(f"== {name} {k} M={M} K={ci if k == 'fwd' else co} N={co if k == 'fwd' else ci} tiles={tiles} launch {us:.1f} us")
        t_first, t_last = s[:, :, 7].min(), (s[:, :, 7] + s[:, :, 5]).max()
        print(f"   first wave start -> last wave end: {(t_last - t_first) / 100.0:.1f} us (s_memrealtime)")
        for role, sl in (("consumer", slice(0, 8)), ("producer", slice(8, 12))):
            r = s[:, sl, :]
            med = lambda i: float(np.median(r[:, :, i]))
            clock = np.median(r[:, :, 6] / np.maximum(r[:, :, 5], 1)) * 100.0
            nks = (ci if k == "fwd" else co) // 32
            print(f"   {role}: prologue {med(0):8.0f}  loop {med(1):8.0f} ({med(1) / nks:6.0f}/step)  epilogue {med(2):8.0f}  barrier-wait {med(3):8.0f} ({med(3) / nks:6.0f}/step)"
                  f"  {'store_a' if role == 'producer' else 'vmcnt(0)'} {med(4):8.0f} ({med(4) / nks:6.0f}/step)  total {med(6):8.0f} cyc  clock {clock:.0f} MHz")
        # spread of the workgroups' start times (rounds): histogram of start offsets in us
        st = (s[:, 0, 7] - t_first) / 100.0
        print("   start offsets (us) quantiles 0/25/50/75/100:", np.round(np.quantile(st, [0, .25, .5, .75, 1.0]), 1))
