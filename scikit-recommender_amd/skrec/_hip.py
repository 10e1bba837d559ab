"""ctypes binding of ``libskrec_hip.so`` (C ABI declared in ``include/skrec_hip.h``).

This is the only way the host mirror reaches the hot path: there is no CPU fallback.  If the
library is missing, or no gfx950 device is visible when a kernel is needed, the call raises.
PyTorch is used for device memory and streams only: tensors go in as ``data_ptr()`` and the
current stream as ``cuda_stream``.
"""
import ctypes as C
import os

import numpy as np

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # scikit-recommender_amd/
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libskrec_hip.so")
CSRC_DIR = os.path.join(_PKG_ROOT, "csrc")

vp, i32, i64, u32, u64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/skrec_hip.h declares
SIGNATURES = {
    "skr_abi_version": (i32, []),
    "skr_last_error": (C.c_char_p, []),
    "skr_device_count": (i32, []),
    "skr_device_summary": (i32, [C.c_char_p, sz]),
    "skr_csr_max_row_len": (i32, [vp, i32, C.POINTER(i32), vp]),
    "skr_sampler_create": (i32, [u32, C.POINTER(vp)]),
    "skr_sampler_destroy": (i32, [vp]),
    "skr_sampler_get_state": (i32, [vp, vp, C.POINTER(i32)]),
    "skr_sampler_set_state": (i32, [vp, vp, i32]),
    "skr_sampler_draws": (i32, [vp, C.POINTER(u64)]),
    "skr_sampler_last_epoch": (i32, [vp, C.POINTER(i64)]),
    "skr_randint_choice": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp]),
    "skr_sample_epoch_exact": (i32, [vp, i32, i32, vp, vp, i64, i32, vp, vp]),
    "skr_sample_epoch_exact_counts": (i32, [vp, i32, i32, vp, vp, i64, vp, i64, vp, vp]),
    "skr_csr_row_stats": (i32, [vp, i32, C.POINTER(i64), vp]),
    "skr_sample_epoch_exact_stats": (i32, [vp, i32, i32, vp, vp, i64, i32, vp, C.POINTER(i64), vp]),
    "skr_sample_epoch_fast": (i32, [u64, u64, i64, i32, i32, vp, vp, i64, i32, vp, vp]),
    "skr_adam_block_mark": (i32, [vp, i64, i64, i32, vp, i32, vp, i64, vp]),
    "skr_adam_block_cold": (i32, [vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, vp, i32, vp]),
    "skr_adam_block_hot": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, i64, vp, i64, i64, i32, vp, vp]),
    "skr_adam_step_tf": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, vp, vp]),
    "skr_adam_block_cold_tf": (i32, [vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, vp, i32, vp]),
    "skr_adam_block_hot_tf": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, i64, vp, i64, i64, i32, vp, vp]),
    "skr_bpr_fused_plan": (i32, [vp, vp, vp, i32, i32, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp]),
    "skr_bpr_fused_step": (i32, [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i32, i64, i64, i64, f32, f32, f32, f32, i64, i32, i32,
                                 f32, vp, vp]),
    "skr_bpr_fused_block": (i32, [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i32, i64, i64, i64, f32, f32, f32, f32, i64, i32, f32, vp,
                                  i64, vp, vp, vp, vp, i32, vp]),
    "skr_bpr_fused_plan2": (i32, [vp, vp, vp, i32, i32, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, i32, vp]),
    "skr_bpr_fused_pre": (i32, [vp, vp, vp, i64, vp, i64, vp, vp, vp, f32, f32, f32, f32, i64, i32, vp, i32, vp]),
    "skr_bpr_fused_step2": (i32, [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i32, i64, i64, i64, f32, f32, f32, f32, i64, i32, i32,
                                  f32, vp, vp, vp]),
    "skr_bpr_fused_block2": (i32, [vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, i32, i64, i64, i64, f32, f32, f32, f32, i64, i32, f32, vp,
                                   i64, vp, vp, vp, vp, i32, vp, vp]),
    "skr_bpr_fused_end": (i32, [vp, vp, vp, i64, vp, i64, vp, vp, vp, f32, f32, f32, f32, i64, i32, vp, i32, i32, vp]),
    "skr_host_permutation": (i32, [vp, C.POINTER(i32), i64, vp]),
    "skr_shuffle_gather": (i32, [vp, u64, i64, i64, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(vp), vp]),
    "skr_shuffle_permutation": (i32, [u64, i64, i64, vp, vp]),
    "skr_shuffle_permutation_host": (i32, [u64, i64, i64, vp]),
    "skr_cold_pass_census": (i32, [C.POINTER(u64), i32]),
    "skr_selftest_cold_math": (i32, [u64, C.POINTER(u64), vp]),
    "skr_pack_grad_rows": (i32, [vp, i32, vp, vp, i32, vp, vp]),
    "skr_unpack_grad_rows": (i32, [vp, i32, i32, vp, vp, i32, vp, vp, vp]),
    "skr_unpack_grad_rows_sorted": (i32, [vp, i32, i32, vp, vp, i32, vp, vp, vp]),
    "skr_gru_cell_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "skr_gru_cell_bwd": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "skr_gru_cell_bwd_scatter": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp,
                                       vp]),
    "skr_session_loss": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp]),
    "skr_session_loss_sharded": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, i32, i32, vp]),
    "skr_session_loss_grads": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp]),
    "skr_session_out_grads": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, f32, vp, vp, vp, vp, vp]),
    "skr_pop_sample": (i32, [vp, i32, vp, u64, i64, vp, vp]),
    "skr_scatter_add_rows": (i32, [vp, vp, i32, i32, vp, f32, vp, vp, vp, vp]),
    "skr_eval_scores": (i32, [vp, i32, i32, i64, vp, vp, C.POINTER(i32), i32, i32, vp, vp, vp, vp]),
    "skr_eval_fused_workspace": (sz, [i32, i32]),
    "skr_eval_fused_rejected": (i32, [vp, vp]),
    "skr_eval_fused_topk": (i32, [vp, vp, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, sz, vp]),
    "skr_mask_train": (i32, [vp, i32, i32, i64, vp, vp, vp, vp]),
    "skr_score_matrix": (i32, [vp, vp, i32, vp, vp, i32, i32, vp, i64, vp]),
    "skr_rank_metrics": (i32, [vp, i32, i32, vp, vp, vp, C.POINTER(i32), i32, vp, vp, vp]),
    "skr_bpr_step": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "skr_bpr_step_sharded": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "skr_bpr_step_spread": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "skr_bpr_step_dim": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp, vp, f32, vp]),
    "skr_csr_spmm_strided": (i32, [i32, vp, vp, vp, vp, i32, i32, i64, vp, vp, vp, f32, vp]),
    "skr_adam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, i32, vp, vp]),
    "skr_csr_spmm": (i32, [i32, vp, vp, vp, vp, i32, i64, vp, vp, vp, f32, vp]),
    "skr_spmm_plan_create": (i32, [i32, i32, vp, vp, vp, i64, i32, C.POINTER(vp), vp]),
    "skr_spmm_plan_run": (i32, [vp, vp, i32, vp, vp, vp, f32, vp]),
    "skr_spmm_plan_run_masked": (i32, [vp, vp, i32, vp, vp, vp, f32, vp, vp, vp]),
    "skr_spmm_plan_run_ex": (i32, [vp, vp, i32, vp, vp, vp, vp]),
    "skr_clear_marked_rows": (i32, [vp, i64, i64, vp, i32, vp]),
    "skr_layer_refine_bwd_masked": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, vp, i32, vp]),
    "skr_mark_ids": (i32, [vp, i64, i64, vp, vp]),
    "skr_csr_scatter_marked_rows": (i32, [i32, vp, vp, vp, vp, vp, i32, vp, vp]),
    "skr_spmm_plan_info": (i32, [vp, C.POINTER(i64)]),
    "skr_spmm_plan_destroy": (i32, [vp]),
    "skr_layer_refine_fwd": (i32, [vp, vp, i64, i32, vp, vp, vp, vp]),
    "skr_layer_refine_bwd": (i32, [vp, vp, vp, vp, i64, i32, vp, vp, vp]),
    "skr_gather_rows": (i32, [vp, vp, i64, i32, vp, vp]),
    "skr_scatter_rows": (i32, [vp, vp, i64, i32, vp, vp]),
    "skr_sum_blocks": (i32, [vp, i32, i64, vp, vp]),
    "skr_axpy": (i32, [f32, vp, vp, i64, vp]),
    "skr_scale_copy": (i32, [f32, vp, vp, i64, vp]),
    "skr_scale": (i32, [f32, vp, i64, vp]),
}

class SpmmEpilogue(C.Structure):
    """skr_spmm_epilogue (include/skrec_hip.h): what happens to a finished row of a propagation"""
    _fields_ = [("mode", C.c_int32), ("accum_init", C.c_int32), ("addend", vp), ("Y", vp), ("accum", vp), ("accum_base", vp),
                ("accum_scale", C.c_float), ("ld", C.c_int32), ("E", vp), ("w", vp), ("Z", vp), ("rawY", vp), ("dE", vp), ("accum_mask", vp), ("addend_mask", vp)]


EPI_PLAIN, EPI_REFINE_FWD, EPI_REFINE_BWD = 0, 1, 2

SKR_MAX_TOPK = 128            # skr_eval_fused_topk
SKR_MAX_TOPK_SCORES = 512     # skr_eval_scores, skr_rank_metrics
SKR_LOSS_SLOTS = 32      # skr_bpr_step_spread: pairs of loss words per batch


class HipError(RuntimeError):
    pass


_lib = None


def build(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-C", CSRC_DIR, "-j4"], check=True)
    return LIB_PATH


def lib():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(there is no CPU fallback for the hot path)")
        # torch first: it ships its own libamdhip64; if this library pulled in /opt/rocm's copy before torch
        # loaded, the process would hold two HIP runtimes and torch.cuda would report no device afterwards
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = the .so and the header disagree
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().skr_last_error()
        raise (ValueError if rc == -1 else HipError)(msg.decode() if msg else f"libskrec_hip error {rc}")


def require_gpu():
    import torch
    if not torch.cuda.is_available() or lib().skr_device_count() == 0:
        raise HipError("no MI355X / HIP device is visible: the skrec hot path runs on the GPU only")
    return torch.device("cuda", torch.cuda.current_device())


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """device (or host) address of a torch tensor / numpy array / None"""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    assert t.is_contiguous(), "tensor must be contiguous"
    return t.data_ptr()


def score_matrix(user_table, users, item_table, bias):
    """predict() helper: dense [len(users), n_items] fp32 scores on the device (skr_score_matrix)"""
    import torch
    dev = user_table.device
    du = torch.as_tensor(np.asarray(users, dtype=np.int32)).to(dev)
    n_items = int(item_table.shape[0])
    out = torch.empty((du.numel(), n_items), dtype=torch.float32, device=dev)
    check(lib().skr_score_matrix(ptr(user_table), ptr(du), du.numel(), ptr(item_table), ptr(bias), n_items,
                                 int(item_table.shape[1]), ptr(out), n_items, stream()))
    return out


def host_permutation(n):
    """np.random.permutation(n) of numpy's GLOBAL legacy generator, as int32, computed natively (skr_host_permutation):
    same values, same generator state afterwards; the GIL is released while it runs"""
    kind, key, pos, has_gauss, cached = np.random.get_state()
    if kind != "MT19937":
        return np.random.permutation(n).astype(np.int32)
    key = np.ascontiguousarray(key, dtype=np.uint32).copy()
    cpos = i32(int(pos))
    out = np.empty(int(n), dtype=np.int32)
    check(lib().skr_host_permutation(key.ctypes.data, C.byref(cpos), int(n), out.ctypes.data))
    np.random.set_state((kind, key, int(cpos.value), has_gauss, cached))
    return out


def shuffle_gather(columns, perm=None, seed=0, n_out=None):
    """One launch for all columns (skr_shuffle_gather): row r of every result = row perm[r] of the column, or row
    pi_seed(r) when ``perm`` is None.  ``columns``: contiguous 4-byte-element device tensors [n] or [n, w]."""
    import torch
    n = int(columns[0].shape[0])
    n_out = n if n_out is None else int(n_out)
    if perm is not None:      # the kernel reads int32 indices and trusts them
        assert perm.dtype == torch.int32 and perm.is_contiguous() and perm.numel() >= n_out, \
            "shuffle_gather: perm must be a contiguous int32 tensor with at least n_out entries"
    outs = []
    for s in range(0, len(columns), 4):
        cols = columns[s:s + 4]
        for c in cols:
            assert c.is_contiguous() and c.element_size() == 4 and c.shape[0] == n
        res = [torch.empty((n_out,) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device) for c in cols]
        k = len(cols)
        check(lib().skr_shuffle_gather(ptr(perm), int(seed), n, n_out, k, (vp * k)(*[c.data_ptr() for c in cols]),
                                       (i32 * k)(*[max(1, c.numel() // max(n, 1)) for c in cols]),
                                       (vp * k)(*[r.data_ptr() for r in res]), stream()))
        outs.extend(res)
    return outs


def metric_array(ids):
    return (i32 * len(ids))(*[int(x) for x in ids])
