"""ctypes binding of libttk_hip.so (C-ABI: include/ttk.h).

There is NO fallback: if the shared object is missing or a symbol is absent, importing the kernels
raises; an entry point that returns non-zero raises RuntimeError(ttk_last_error_string()).
"""
from __future__ import annotations

import functools

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTK_LIB") or os.path.join(os.path.dirname(_HERE), "libttk_hip.so")  # TTK_LIB: A/B builds

_P, _I, _L, _F, _D = c_void_p, c_int, c_int64, c_float, c_double

# name -> argument types (the trailing stream pointer is added automatically)
_SIGNATURES = {
    "ttk_bn_fwd_finalize": [_P, _P, _I, _I, _L, _P, _P, _P, _P, _P, _F, _F, _P],
    "ttk_bn_bwd_frozen": [_P, _I],
    "ttk_bn_frozen_bound": [_P, _P, _I, _I, _L, _P],
    "ttk_bn_eval_prepare": [_P, _P, _P, _P, _F, _I, _P],
    "ttk_bn_bwd_finalize": [_P, _I, _I, _L, _P, _P, _P, _P, _I],
    "ttk_stem_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I],
    "ttk_stem_bwd_weight": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I],
    "ttk_dwconv3x3_fwd": [_P] * 8 + [_I] * 6,
    "ttk_dwconv3x3_bwd_data": [_P] * 12 + [_I, _P] + [_I] * 6,
    "ttk_pwconv1x1_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _I],
    "ttk_pwconv1x1_bwd_data": [_P] * 8 + [_L, _I, _I, _P, _I],
    "ttk_pwconv1x1_bwd_weight": [_P] * 7 + [_L, _I, _I, _I],
    "ttk_pwconv_prepare_weights": [_I, _P, _P, _P, _P],
    "ttk_pwconv1x1_bwd_fused": [_P] * 11 + [_L, _I, _I],
    "ttk_transpose": [_P, _P, _I, _I],
    "ttk_avgpool_fwd": [_P, _P, _P, _P, _I, _I, _I, _I],
    "ttk_avgpool_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I],
    "ttk_bn_act": [_P, _P, _P, _P, _L, _I],
    "ttk_stem7_fwd": [_P, _P, _P, _P, _P, _I, _I, _I],
    "ttk_stem7_bwd_weight": [_P, _P, _P, _P, _P, _P, _I, _I, _I],
    "ttk_maxpool3x3s2_fwd": [_P, _P, _P, _P, _I, _I, _I, _I],
    "ttk_maxpool3x3s2_bwd": [_P] * 7 + [_I] * 4,
    "ttk_bn_add_act": [_P, _P, _P, _P, _P, _P, _I, _L, _I],
    "ttk_residual_bwd": [_P] * 10 + [_L, _I],
    "ttk_bn_bwd_apply": [_P, _P, _P, _P, _L, _I],
    "ttk_conv_weight_repack": [_P, _P, _P, _I, _I, _I, _I],
    "ttk_conv_prepare_weights": [_I, _P, _P, _P, _P, _P, _P],
    "ttk_conv_fwd": [_P, _P, _P, _P, _P, _P] + [_I] * 9,
    "ttk_conv_bwd_data": [_P] * 8 + [_I] * 9,
    "ttk_conv_bwd_weight": [_P] * 7 + [_I] * 9,
    "ttk_heads_fwd": [_P] * 8 + [_I] * 7 + [_P] * 9,
    "ttk_heads_bwd": [_P] * 8 + [_I] * 7 + [_P] * 15,
    "ttk_diag_scale_fwd": [_P, _P, _I],
    "ttk_diag_scale_bwd": [_P, _P, _P, _I],
    "ttk_loss_rot_fwd": [_P, _P, _I, _P],
    "ttk_loss_rot_bwd": [_P, _P, _P, _I, _P],
    "ttk_loss_rot6d_fwd": [_P, _P, _I, _P],
    "ttk_loss_rot6d_bwd": [_P, _P, _I, _P],
    "ttk_loss_ortho6d_fwd": [_P, _I, _P],
    "ttk_loss_ortho6d_bwd": [_P, _P, _I, _P],
    "ttk_mat_to_quat_fwd": [_P, _I, _P],
    "ttk_mat_to_quat_bwd": [_P, _P, _I, _P],
    "ttk_loss_quatreg_fwd": [_P, _I, _P],
    "ttk_loss_quatreg_bwd": [_P, _P, _I, _P],
    "ttk_loss_mse_rows_fwd": [_P, _P, _I, _I, _P],
    "ttk_loss_mse_rows_bwd": [_P, _P, _P, _I, _I, _P],
    "ttk_loss_mse_cols_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "ttk_loss_mse_cols_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "ttk_multi_copy": [_I, _P, _P, _P],
    "ttk_weighted_sum_fwd": [_I, _P, _P, _P, _P, _F, _P],
    "ttk_weighted_sum_bwd": [_I, _P, _P, _P, _P, _F, _P],
    "ttk_loss_points_fwd": [_P, _P, _I, _I, _F, _F, _P],
    "ttk_loss_points_bwd": [_P, _P, _P, _I, _I, _F, _F, _P],
    "ttk_loss_nllrot_fwd": [_P, _P, _P, _I, _P],
    "ttk_loss_nllrot_bwd": [_P, _P, _P, _P, _I, _P, _P],
    "ttk_loss_nllcoord_fwd": [_P, _P, _P, _I, _P],
    "ttk_loss_nllcoord_bwd": [_P, _P, _P, _P, _I, _P, _P],
    "ttk_loss_normal_fwd": [_P, _P, _P, _I, _I, _I, _I, _F, _F, _P],
    "ttk_loss_normal_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P],
    "ttk_loss_laplace_fwd": [_P, _P, _P, _I, _I, _I, _I, _F, _F, _P],
    "ttk_loss_laplace_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P],
    "ttk_loss_elem_fwd": [_P, _P, _P, _I, _I, _I, _F, _P],
    "ttk_loss_elem_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "ttk_loss_rot_geodesic_fwd": [_P, _P, _I, _P],
    "ttk_loss_rot_geodesic_bwd": [_P, _P, _P, _I, _P],
    "ttk_loss_gmm_fwd": [_P, _P, _P, _P, _I, _D, _I, _P, _P],
    "ttk_loss_gmm_bwd": [_P, _P, _P, _P, _I, _D, _P, _I, _P],
    "ttk_loss_batch": [_I, _P],
    "ttk_blur3x3_fwd": [_P, _P, _I, _I, _I, _I, _I],
    "ttk_blur3x3_bwd": [_P, _P, _P, _I, _I, _I, _I, _I],
    "ttk_view_roi": [_P, _P, _P, _F, _I, _P],
    "ttk_roi_transform": [_P, _P, _I, _I, _P],
    "ttk_affine_warp": [_P, _I, _I, _I, _I, _P, _P, _I, _F, _F],
    "ttk_affine_labels": [_P, _I, _I, _P, _P, _P, _P, _P],
    "ttk_intensity_augment": [_P, _P, _P, _P, _I, _I, _I, _F],
    "ttk_clip_adam": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "ttk_stream_probe": [_P, _P, _P, _L, _I, _I, _I, _I, _L, _I, _I, _I],
    # bf16-compute path (csrc/bc_*.hip)
    "ttk_bc_prepare_weights": [_I, _P, _P, _P, _P],
    "ttk_bc_pw_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _I],
    "ttk_bc_pw_bwd_data": [_P] * 8 + [_L, _I, _I],
    "ttk_bc_pw_bwd_weight": [_P] * 7 + [_L, _I, _I],
    "ttk_bc_pw_bwd_fused": [_P] * 10 + [_L, _I, _I],
    "ttk_bc_dw_fwd": [_P] * 8 + [_I] * 5,
    "ttk_bc_avgpool_fwd": [_P, _P, _P, _P, _I, _I, _I],
    "ttk_bc_avgpool_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I],
    "ttk_bc_dw_bwd_data": [_P] * 12 + [_I, _P] + [_I] * 5,
    "ttk_bc_bn_bwd_finalize_fold": [_P, _I, _I, _L, _P, _P, _P, _P, _I, _P, _I, _L, _P, _I],
}

ABI_VERSION = 28


# Whether the backbones hand the running mean to the forward producers as the statistics pivot (include/ttk.h).  Always on in the
# product; tests/test_bn_pivot_gpu.py and tools flip this attribute for the A/B against plain sums of y and y^2.
BN_PIVOT = True


def bn_pivot() -> bool:
    return BN_PIVOT


class LossOp(ctypes.Structure):
    """ttk_loss_op (include/ttk.h): one loss op of a ttk_loss_batch launch."""
    _fields_ = [("kind", c_int), ("items", c_int), ("p", c_void_p * 6), ("i", c_int * 4), ("f", c_float * 2), ("d", c_double)]


LOSS_BATCH_MAX = 32
# entry point -> (TTK_OP_* kind, threads the op needs as a function of its argument tuple); the arguments are packed into
# p[] / i[] / f[] / d in signature order
_n = lambda k: (lambda a: a[k])
LOSS_BATCH_OPS = {
    "ttk_loss_rot_fwd": (0, _n(2)), "ttk_loss_rot_bwd": (1, _n(3)), "ttk_loss_rot6d_fwd": (2, _n(2)), "ttk_loss_rot6d_bwd": (3, _n(2)),
    "ttk_loss_ortho6d_fwd": (4, _n(1)), "ttk_loss_ortho6d_bwd": (5, _n(2)), "ttk_loss_quatreg_fwd": (6, _n(1)), "ttk_loss_quatreg_bwd": (7, _n(2)),
    "ttk_loss_mse_rows_fwd": (8, lambda a: a[2] * 64), "ttk_loss_mse_rows_bwd": (9, lambda a: a[3] * a[4]),
    "ttk_loss_mse_cols_fwd": (10, lambda a: a[2] * 64), "ttk_loss_mse_cols_bwd": (11, lambda a: a[3] * a[4]),
    "ttk_loss_points_fwd": (12, lambda a: a[2] * 64), "ttk_loss_points_bwd": (13, lambda a: a[3] * 204),
    "ttk_loss_nllrot_fwd": (14, _n(3)), "ttk_loss_nllrot_bwd": (15, _n(4)), "ttk_loss_nllcoord_fwd": (16, _n(3)), "ttk_loss_nllcoord_bwd": (17, _n(4)),
    "ttk_loss_normal_fwd": (18, lambda a: a[3] * 64), "ttk_loss_normal_bwd": (19, lambda a: a[4] * (204 if a[6] else a[5])),
    "ttk_loss_gmm_fwd": (20, lambda a: a[6] * 64), "ttk_loss_gmm_bwd": (21, lambda a: a[7] * 50),
}


class _Library:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP kernels are not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "neuralnet-tracker-traincode_amd/csrc`). There is no CPU/PyTorch fallback for the training path."
            )
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.cdll.ttk_last_error_string.restype = c_char_p
        self.cdll.ttk_abi_version.restype = c_int
        self.cdll.ttk_clear_error.restype = c_int
        self._clear = self.cdll.ttk_clear_error
        v = self.cdll.ttk_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f"libttk_hip.so ABI version {v}, host expects {ABI_VERSION}: rebuild")
        for name in ("ttk_partial_rows_elementwise", "ttk_partial_rows_gemm"):
            fn = getattr(self.cdll, name)
            fn.argtypes, fn.restype = [c_int64], c_int
        self.cdll.ttk_partial_rows_pwconv.argtypes, self.cdll.ttk_partial_rows_pwconv.restype = [c_int64, c_int, c_int, c_int], c_int
        self.cdll.ttk_pwconv_tile_rows.argtypes, self.cdll.ttk_pwconv_tile_rows.restype = [c_int64, c_int, c_int, c_int], c_int
        self.cdll.ttk_heads_num_rows.argtypes, self.cdll.ttk_heads_num_rows.restype = [c_int, c_int, c_int], c_int
        self.cdll.ttk_pwconv_prepared_bytes.argtypes, self.cdll.ttk_pwconv_prepared_bytes.restype = [c_int, c_int], ctypes.c_size_t
        self.cdll.ttk_partial_rows_dwconv.argtypes, self.cdll.ttk_partial_rows_dwconv.restype = [c_int] * 6, c_int
        self.cdll.ttk_pwconv_wgrad_partial_bytes.argtypes, self.cdll.ttk_pwconv_wgrad_partial_bytes.restype = [c_int64, c_int, c_int], ctypes.c_size_t
        self.cdll.ttk_pwconv_wgrad_scratch_bytes.argtypes, self.cdll.ttk_pwconv_wgrad_scratch_bytes.restype = [c_int64, c_int, c_int], ctypes.c_size_t
        self.cdll.ttk_conv_wgrad_partial_bytes.argtypes, self.cdll.ttk_conv_wgrad_partial_bytes.restype = [c_int] * 9, ctypes.c_size_t
        self.cdll.ttk_pwconv1x1_bwd_fused_rows.argtypes, self.cdll.ttk_pwconv1x1_bwd_fused_rows.restype = [c_int64, c_int, c_int], c_int
        self.cdll.ttk_pwconv1x1_bwd_fused_partial_bytes.argtypes, self.cdll.ttk_pwconv1x1_bwd_fused_partial_bytes.restype = [c_int64, c_int, c_int], ctypes.c_size_t
        self.cdll.ttk_stem7_wgrad_partial_bytes.argtypes, self.cdll.ttk_stem7_wgrad_partial_bytes.restype = [c_int] * 3, ctypes.c_size_t
        self.cdll.ttk_stem_wgrad_partial_bytes.argtypes, self.cdll.ttk_stem_wgrad_partial_bytes.restype = [], ctypes.c_size_t
        self.cdll.ttk_bc_prepared_bytes.argtypes, self.cdll.ttk_bc_prepared_bytes.restype = [c_int, c_int], ctypes.c_size_t
        self.cdll.ttk_bc_partial_rows_pw.argtypes, self.cdll.ttk_bc_partial_rows_pw.restype = [c_int64, c_int, c_int], c_int
        self.cdll.ttk_bc_partial_rows_dw.argtypes, self.cdll.ttk_bc_partial_rows_dw.restype = [c_int] * 6, c_int
        self.cdll.ttk_bc_partial_rows_pool.argtypes, self.cdll.ttk_bc_partial_rows_pool.restype = [c_int] * 3, c_int
        self.cdll.ttk_bc_pw_wgrad_scratch_bytes.argtypes, self.cdll.ttk_bc_pw_wgrad_scratch_bytes.restype = [c_int64, c_int, c_int], ctypes.c_size_t
        self.cdll.ttk_bc_pw_bwd_fused_rows.argtypes, self.cdll.ttk_bc_pw_bwd_fused_rows.restype = [c_int64, c_int, c_int], c_int
        self.cdll.ttk_bc_pw_wgrad_slices.argtypes, self.cdll.ttk_bc_pw_wgrad_slices.restype = [c_int64, c_int, c_int], c_int
        self.cdll.ttk_bc_pw_bwd_fused_scratch_bytes.argtypes, self.cdll.ttk_bc_pw_bwd_fused_scratch_bytes.restype = [c_int64, c_int, c_int], ctypes.c_size_t
        self._fns = {}
        self._stale_reported = False
        for name, sig in _SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the symbol is missing: loud by design
            fn.argtypes = list(sig) + [c_void_p]
            fn.restype = c_int
            self._fns[name] = fn

    def clear_stale_error(self, where: str = "") -> int:
        """Drops (and reports once) a HIP error that an EARLIER, unrelated call of the process left pending - entry points report launch
        failures through the sticky per-thread hipGetLastError().  Called when the library is loaded and once per training / evaluation
        step (train.training_step, eval.Predictor), not before every launch: that was a second ctypes round trip in front of each of the
        ~160 launches of a step (round 3: host_enqueue_ms_per_step 3.2)."""
        stale = self._clear()
        if stale and not self._stale_reported:
            self._stale_reported = True
            import warnings
            warnings.warn(f"a HIP error (hipError_t {stale}) from an earlier launch was pending{' at ' + where if where else ''}; it was dropped, "
                          "not raised", RuntimeWarning, stacklevel=2)
        return stale

    def call(self, name: str, *args):
        # (the raw handle of torch's current stream on the current device: two C calls.  `torch.cuda.current_stream().cuda_stream` builds a Stream
        # object and resolves the device index in Python - 10 us per launch, 1.5 ms of the ~160 launches of a step until round 5)
        rc = self._fns[name](*args, _raw_stream(_cur_device()))
        if rc != 0:
            msg = self.cdll.ttk_last_error_string().decode(errors="replace")
            hint = "" if rc < 0 else " (a positive code is a hipError_t read from the sticky hipGetLastError(): an earlier launch of this step may have left it)"
            raise RuntimeError(f"{name} failed (code {rc}): {msg}{hint}")

    def loss_batch(self, ops):
        """ttk_loss_batch: `ops` = [(entry point name, argument tuple as for call()), ...], mutually independent; one launch
        per LOSS_BATCH_MAX ops."""
        for lo in range(0, len(ops), LOSS_BATCH_MAX):
            chunk = ops[lo:lo + LOSS_BATCH_MAX]
            arr = (LossOp * len(chunk))()
            for o, (name, args) in zip(arr, chunk):
                kind, items = LOSS_BATCH_OPS[name]
                o.kind, o.items = kind, int(items(args))
                np_ = ni = nf = 0
                for ty, v in zip(_SIGNATURES[name], args):
                    if ty is _P:
                        o.p[np_] = v
                        np_ += 1
                    elif ty is _I:
                        o.i[ni] = int(v)
                        ni += 1
                    elif ty is _F:
                        o.f[nf] = float(v)
                        nf += 1
                    else:
                        o.d = float(v)
            self.call("ttk_loss_batch", len(chunk), arr)

    def pwconv_prepared_bytes(self, cin: int, cout: int) -> int:
        return self.cdll.ttk_pwconv_prepared_bytes(cin, cout)

    def conv_wgrad_partial_bytes(self, B, H, W, cin, cout, k, stride) -> int:
        """Scratch bytes of ttk_conv_bwd_weight's slice-wise (atomic-free) weight gradient; 0 = the call takes none."""
        return self.cdll.ttk_conv_wgrad_partial_bytes(B, H, W, cin, cout, k, k, stride, k // 2)

    def pwconv_wgrad_partial_bytes(self, m: int, cin: int, cout: int) -> int:
        """Scratch bytes of the deterministic (fixed-order) weight-gradient reduction; 0 = this shape has none."""
        return self.cdll.ttk_pwconv_wgrad_partial_bytes(m, cin, cout)

    @functools.lru_cache(maxsize=None)
    def pwconv_wgrad_scratch_bytes(self, m: int, cin: int, cout: int) -> int:
        """Scratch bytes ttk_pwconv1x1_bwd_weight wants as `partial` in the DEFAULT mode (0: the shape runs its atomic form)."""
        return self.cdll.ttk_pwconv_wgrad_scratch_bytes(m, cin, cout)

    def conv_prepare_weights(self, weights, w_fwd, w_bwd):
        """ttk_conv_prepare_weights: `weights[i]` [Cout, Cin, k, k] fp32, `w_fwd[i]` / `w_bwd[i]` int16 buffers of 3 * numel
        elements (either may be None)."""
        n = len(weights)
        PA, IA = c_void_p * n, c_int * n
        self.call("ttk_conv_prepare_weights", n, PA(*[ptr(w) for w in weights]), PA(*[ptr(t) for t in w_fwd]), PA(*[ptr(t) for t in w_bwd]),
                  IA(*[int(w.shape[0]) for w in weights]), IA(*[int(w.shape[1]) for w in weights]), IA(*[int(w.shape[2]) for w in weights]))

    def pwconv_prepare_weights(self, weights, prepared):
        """One launch: forward and data-gradient weight operands of every pointwise layer (`weights[i]`: [Cout, Cin(,1,1)]
        fp32, `prepared[i]`: uint8 scratch of pwconv_prepared_bytes)."""
        n = len(weights)
        wp = (c_void_p * n)(*[ptr(w) for w in weights])
        pp = (c_void_p * n)(*[ptr(q) for q in prepared])
        ci = (c_int * n)(*[int(w.shape[1]) for w in weights])
        co = (c_int * n)(*[int(w.shape[0]) for w in weights])
        self.call("ttk_pwconv_prepare_weights", n, wp, ci, co, pp)

    def bc_prepare_weights(self, weights, prepared):
        """ttk_bc_prepare_weights: the bf16 weight images (forward + data gradient) of every pointwise layer, one launch."""
        n = len(weights)
        wp = (c_void_p * n)(*[ptr(w) for w in weights])
        pp = (c_void_p * n)(*[ptr(q) for q in prepared])
        ci = (c_int * n)(*[int(w.shape[1]) for w in weights])
        co = (c_int * n)(*[int(w.shape[0]) for w in weights])
        self.call("ttk_bc_prepare_weights", n, wp, ci, co, pp)

    def multi_copy(self, srcs, dsts):
        """dsts[k].copy_(srcs[k]) (None: zero fill) for up to 32 contiguous float32 tensors per launch."""
        for i in range(0, len(dsts), 32):
            s, d = srcs[i:i + 32], dsts[i:i + 32]
            n = len(d)
            for a, b in zip(s, d):
                if b.dtype != torch.float32 or not b.is_contiguous() or (a is not None and (a.dtype != torch.float32 or not a.is_contiguous() or a.numel() != b.numel())):
                    raise RuntimeError("multi_copy: contiguous float32 tensors of matching sizes expected")
            self.call("ttk_multi_copy", n, (c_void_p * n)(*[ptr(a) for a in s]), (c_void_p * n)(*[ptr(b) for b in d]),
                      (c_int64 * n)(*[b.numel() for b in d]))

    # (pure functions of their integer arguments, asked ~80 times per step: cached)
    @functools.lru_cache(maxsize=None)
    def partial_rows_elementwise(self, items: int) -> int:
        return self.cdll.ttk_partial_rows_elementwise(items)

    @functools.lru_cache(maxsize=None)
    def partial_rows_dwconv(self, B, H, W, C, stride, backward) -> int:
        return self.cdll.ttk_partial_rows_dwconv(B, H, W, C, stride, int(backward))

    @functools.lru_cache(maxsize=None)
    def partial_rows_gemm(self, m: int, k: int | None = None, nout: int | None = None, dgrad: bool = False) -> int:
        """Rows of BatchNorm partial sums a GEMM epilogue writes for m rows.  With (k, nout): of ttk_pwconv1x1_fwd (k = Cin, nout = Cout) /
        ttk_pwconv1x1_bwd_data (k = Cout, nout = Cin, dgrad=True), whose tiling depends on the shape; without: the 128-row form (convolutions)."""
        if k is None:
            return self.cdll.ttk_partial_rows_gemm(m)
        return self.cdll.ttk_partial_rows_pwconv(m, k, nout, int(dgrad))


# hipGraph captures run in "global" error mode: a HIP call that another thread makes while a capture is open (an allocation, a copy on
# another stream) invalidates it.  Whoever captures holds this lock (train.GraphedTrainStep); background threads that touch the GPU
# (datasets.resident's host-frame prefetch) take it around their GPU work.
import threading

CAPTURE_LOCK = threading.RLock()

_lib: _Library | None = None


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def lib() -> _Library:
    global _lib
    if _lib is None:
        _lib = _Library()
        if torch.cuda.is_available():
            _lib.clear_stale_error("library load")
    return _lib


def ptr(t: torch.Tensor | None):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("HIP entry point received a non-CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError("HIP entry point received a non-contiguous tensor")
    return t.data_ptr()


def check_tensors(ts, what="tensor"):
    """The checks of ptr() for a list of tensors at once (the backbones validate their parameters and inputs ONCE per call and then
    take the addresses of those and of the tensors they allocate themselves with fast_ptr)."""
    for t in ts:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError(f"HIP entry point received a non-CUDA or non-contiguous {what}")


def fast_ptr(t: torch.Tensor | None):
    """Device pointer without ptr()'s checks: for tensors the caller allocated itself or has run through check_tensors."""
    return None if t is None else t.data_ptr()


CHANNEL_BLOCK = 32  # kCB of csrc/ttk_common.h


def to_blocks(t: torch.Tensor) -> torch.Tensor:
    """Channels-last values [..., C] -> the storage order of the MobileNet kernels' activation tensors (include/ttk.h, "Activation
    layout": channel blocks [C/32][pixels][32]).  The result keeps the nominal shape of `t`; only its memory order differs.  Host code
    never needs this - activations are produced and consumed by kernels - tests and tools do."""
    C = t.shape[-1]
    if C <= CHANNEL_BLOCK:
        return t.contiguous()
    return t.reshape(-1, C // CHANNEL_BLOCK, CHANNEL_BLOCK).transpose(0, 1).contiguous().view(t.shape)


def from_blocks(t: torch.Tensor) -> torch.Tensor:
    """Inverse of `to_blocks`: a tensor whose memory holds channel blocks -> the channels-last values, same nominal shape."""
    C = t.shape[-1]
    if C <= CHANNEL_BLOCK:
        return t
    return t.reshape(C // CHANNEL_BLOCK, -1, CHANNEL_BLOCK).transpose(0, 1).reshape(t.shape)


CHANNEL_BLOCK_BC = 64  # bf16-compute path (csrc/bc_common.h)


def to_blocks64(t: torch.Tensor) -> torch.Tensor:
    """Channels-last values [..., C] -> the bf16-compute path's storage order (channel blocks [C/64][pixels][64]); tests and tools only."""
    C = t.shape[-1]
    if C <= CHANNEL_BLOCK_BC:
        return t.contiguous()
    return t.reshape(-1, C // CHANNEL_BLOCK_BC, CHANNEL_BLOCK_BC).transpose(0, 1).contiguous().view(t.shape)


def from_blocks64(t: torch.Tensor) -> torch.Tensor:
    C = t.shape[-1]
    if C <= CHANNEL_BLOCK_BC:
        return t
    return t.reshape(C // CHANNEL_BLOCK_BC, -1, CHANNEL_BLOCK_BC).transpose(0, 1).reshape(t.shape)


def exported_symbols() -> list[str]:
    return ["ttk_abi_version", "ttk_last_error_string", "ttk_clear_error", "ttk_partial_rows_elementwise",
            "ttk_partial_rows_gemm", "ttk_partial_rows_pwconv", "ttk_pwconv_tile_rows", "ttk_partial_rows_dwconv", "ttk_heads_num_rows", "ttk_pwconv_prepared_bytes",
            "ttk_pwconv_wgrad_partial_bytes", "ttk_pwconv_wgrad_scratch_bytes", "ttk_stem_wgrad_partial_bytes", "ttk_conv_wgrad_partial_bytes", "ttk_stem7_wgrad_partial_bytes",
            "ttk_pwconv1x1_bwd_fused_rows", "ttk_pwconv1x1_bwd_fused_partial_bytes", "ttk_bc_prepared_bytes", "ttk_bc_partial_rows_pw",
            "ttk_bc_partial_rows_dw", "ttk_bc_partial_rows_pool", "ttk_bc_pw_wgrad_scratch_bytes", "ttk_bc_pw_bwd_fused_rows",
            "ttk_bc_pw_bwd_fused_scratch_bytes", "ttk_bc_pw_wgrad_slices"] + list(_SIGNATURES)
