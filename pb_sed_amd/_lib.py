"""ctypes binding of ``libpbsed_mi355.so`` (the C-ABI declared in ``include/pbsed.h``).

The product path has NO fallback: if the shared library is missing or an entry point fails the
caller gets an exception - never a silent eager/CPU path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PBSED_LIB') or os.path.join(_HERE, 'libpbsed_mi355.so')     # PBSED_LIB: a tools/ build variant

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i = C.POINTER(C.c_int)
_u8 = C.POINTER(C.c_ubyte)
_v = C.c_void_p
_pp = C.POINTER(C.c_void_p)
I, F32, F64, SZ = C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> argtypes (all return int status unless listed in _NON_STATUS)
SIGNATURES = {
    'pbsed_last_error': [],
    'pbsed_version': [],
    'pbsed_mix_clips': [_v, _v, _v, _v, I, I, I, _v],
    'pbsed_encode_targets': [_v, _v, _v, _v, _v, _v, _v, I, I, I, _v],
    'pbsed_conv_pack_dims': [I, I, I, I, I, _i, _i],
    'pbsed_pack_conv_weights': [_v, _v, I, I, I, I, I, _v],
    'pbsed_conv_fwd': [_v, _v, _v, _v, _v, I, _v, _v, _v, _v, I, I, I, I, I, I, I, I, I, _v],
    'pbsed_conv_fwd_res': [_v, _v, _v, _v, _v, I, _v, _v, _v, _v, I, I, I, I, I, I, I, I, I, _v, _v],
    'pbsed_pool21_fwd': [_v, _v, _v, SZ, I, _v],
    'pbsed_pool21_bwd_add': [_v, _v, _v, SZ, I, _v],
    'pbsed_add_inplace': [_v, _v, SZ, _v],
    'pbsed_conv_bwd_data': [_v, _v, _v, _v, _v, _v, _v, _v, _v, _v, I, _v, I, I, I, I, I, I, I, _v],
    'pbsed_pack_conv_weights_batched': [_v, I, _v],
    'pbsed_conv_pack_dims_wino': [I, I, I, _i, _i],
    'pbsed_pack_conv_weights_wino': [_v, _v, I, I, I, _v],
    'pbsed_conv_fwd_wino': [_v, _v, _v, _v, _v, I, _v, _v, _v, _v, I, I, I, I, I, I, I, _v],
    'pbsed_conv_bwd_data_wino': [_v, _v, _v, _v, _v, _v, _v, _v, _v, _v, I, _v, I, I, I, I, I, _v],
    'pbsed_conv_pack_dims_winox3': [I, I, I, _i, _i],
    'pbsed_conv_pack_dims_s16': [I, I, I, _i, _i],
    'pbsed_pack_conv_weights_winox3': [_v, _v, I, I, I, _v],
    'pbsed_pack_conv_weights_s16': [_v, _v, I, I, I, _v],
    'pbsed_conv_fwd_winox3': [_v, _v, _v, _v, _v, I, _v, _v, _v, _v, I, I, I, I, I, I, I, _v],
    'pbsed_conv_fwd_s16': [_v, _v, _v, _v, _v, I, _v, _v, _v, _v, I, I, I, I, I, I, I, _v],
    'pbsed_conv_bwd_data_winox3': [_v, _v, _v, _v, _v, _v, _v, _v, _v, _v, I, _v, I, I, I, I, I, _v],
    'pbsed_conv_bwd_data_s16': [_v, _v, _v, _v, _v, _v, _v, _v, _v, _v, I, _v, I, I, I, I, I, _v],
    'pbsed_conv1d_pack_dims_x3': [I, I, I, _i, _i],
    'pbsed_pack_conv1d_weights_x3': [_v, _v, I, I, I, I, _v],
    'pbsed_conv1d_fwd_x3': [_v, _v, _v, _v, _v, I, _v, _v, _v, I, I, I, I, I, _v],
    'pbsed_conv1d_bwd_data_x3': [_v, _v, _v, _v, _v, _v, _v, _v, _v, I, _v, I, I, I, I, I, _v],
    'pbsed_conv_pack_dims_bf16': [I, I, I, _i, _i],
    'pbsed_pack_conv_weights_bf16': [_v, _v, I, I, I, I, I, I, _v],
    'pbsed_conv_fwd_bf16': [_v, _v, _v, _v, _v, I, _v, _v, _v, _v, I, I, I, I, I, I, I, I, I, I, _v],
    'pbsed_conv_fwd_bf16_res': [_v, _v, _v, _v, _v, I, _v, _v, _v, _v, I, I, I, I, I, I, I, I, I, I, _v, _v],
    'pbsed_conv_bwd_data_bf16': [_v, _v, _v, _v, _v, _v, _v, _v, _v, _v, I, _v, I, I, I, I, I, I, I, I, _v],
    'pbsed_conv_bwd_weight': [_v, _v, _v, I, _v, _v, _v, _v, _v, I, I, I, I, I, I, I, _v],
    'pbsed_conv_bwd_weight_bf16': [_v, _v, _v, I, _v, _v, _v, _v, _v, I, I, I, I, I, I, I, _v],
    'pbsed_conv_bwd_weight_bng': [_v, _v, _v, I, _v, _v, _v, _v, I, _v, _v, _v, _v, _v, I, I, I, I, I, I, I, _v],
    'pbsed_conv_bwd_weight_bng_supported': [I, I, I, I, I, I, I],
    'pbsed_bn_bwd_coef': [_v, F64, _v, _v, _v, _v, _v, _v, I, _v],
    'pbsed_bn_finalize': [_v, F64, _v, _v, F32, F32, _v, _v, _v, _v, _v, _v, I, _v],
    'pbsed_bn_eval_params': [_v, _v, F32, _v, _v, _v, _v, _v, _v, I, _v],
    'pbsed_bn_bwd': [_v, _v, _v, F64, _v, _v, _v, _v, _v, _v, I, I, I, I, _v],
    'pbsed_bn_bwd_finalize': [_v, F64, _v, _v, _v, _v, I, _v],
    'pbsed_bn_bwd_apply': [_v, _v, _v, _v, _v, _v, _v, _v, I, I, I, I, _v],
    'pbsed_augment_logmel': [_v, _v, _v, _v, _v, _v, _v, F32, I, I, I, _v],
    'pbsed_logmel_fwd': [_v, I, I, I, _v, _v, _v, _v, _v, _v, _v, I, I, _v, _v, F32, F32, _v, _v, I, _v, _v],
    'pbsed_logmel_fwd_frames': [_v, I, I, I, _v, _v, _v, _v, _v, _v, _v, _v, I, I, _v, _v, F32, F32, _v, _v, _v, _v],
    'pbsed_logmel_from_stft': [_v, I, I, I, _v, _v, _v, _v, _v, I, I, _v, _v, F32, F32, _v, _v, _v, _v],
    'pbsed_feature_norm_update': [_v, F64, _v, _v, _v, F32, _v, _v, I, _v],
    'pbsed_bct_to_tbc': [_v, _v, I, I, I, _v],
    'pbsed_tbc_to_bct': [_v, _v, I, I, I, I, _v],
    'pbsed_transpose2d': [_v, _v, I, I, _v],
    'pbsed_gru_wgrad': [I, _pp, _pp, _i, _pp, _pp, I, I, I, I, _v],
    'pbsed_bn_relu_fwd': [_v, _v, _v, _v, _v, I, I, I, I, I, _v],
    'pbsed_bn_relu_bwd': [_v, _v, _v, _v, _v, _v, _v, _v, _v, I, I, I, I, I, _v],
    'pbsed_channel_stats': [_v, _v, _v, I, I, I, I, _v],
    'pbsed_tm_gemm': [I, _pp, _pp, _i, _v, _v, I, I, I, _v],
    'pbsed_gru_wgrad_multi': [I, _pp, _pp, _i, _pp, _pp, I, I, I, _i, I, _v],
    'pbsed_gru_scan_fwd': [I, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v],
    'pbsed_gru_scan_bwd': [I, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v],
    'pbsed_gru_stack_fwd': [I, I, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v],
    'pbsed_gru_stack_bwd_granule': [I, I, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v, C.c_uint, _v, _v],
    'pbsed_gru_stack_bwd_granule_bf16': [I, I, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v, C.c_uint, _v, _v],
    'pbsed_gru_stack_fwd_granule': [I, I, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v, C.c_uint, _v, _v],
    'pbsed_gru_stack_fwd_granule_bf16': [I, I, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v, C.c_uint, _v, _v],
    'pbsed_gru_stack_bwd': [I, I, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _pp, _i, _v, I, I, I, _v],
    'pbsed_fbcrnn_loss': [_v, _v, _v, _v, _v, _v, _v, _v, _v, _v, _v, I, I, I, F32, F32, I, F32, I, _v, _v],
    'pbsed_bicrnn_loss': [_v, _v, _v, _v, _v, _v, _v, I, I, I, I, _v],
    'pbsed_bicrnn_review_summary': [_v, _v, _v, _v, _v, _v, _v, I, I, I, I, _v],
    'pbsed_squash_fwd': [_v, _v, SZ, F32, _v],
    'pbsed_squash_bwd': [_v, _v, _v, SZ, F32, _v],
    'pbsed_ensemble_mean_mask': [_pp, I, _v, _v, I, I, I, _v],
    'pbsed_medfilt': [_v, _v, _v, I, I, _v],
    'pbsed_boundariesfilt': [_v, _v, _v, _v, I, I, _v],
    'pbsed_event_frames': [_v, _v, _v, _v, _v, I, I, I, _v],
    'pbsed_grad_sumsq': [_v, SZ, _v, _v],
    'pbsed_adam_step': [_v, _v, _v, _v, SZ, F32, F32, F32, F32, I, F32, F32, _v, _v, _v, I, _v],
    'pbsed_gru_granule_capacity': [I, I, I, I],
    'pbsed_gru_get_poll_delays': [I, _v],
    'pbsed_gru_set_poll_delays': [I, I, I, I, I],
    'pbsed_gru_set_prof': [_v, I],
    'pbsed_gru_set_xcd_local': [I],
    'pbsed_set_scratch': [_v, SZ, _v],
    'pbsed_set_launch_cus': [I],
    'pbsed_scratch_bytes': [],
    'pbsed_memset_async': [_v, I, SZ, _v],
    'pbsed_comm_id_bytes': [],
    'pbsed_comm_unique_id': [_v],
    'pbsed_comm_create': [_v, I, I, _pp],
    'pbsed_comm_destroy': [_v],
    'pbsed_allreduce_begin': [_v, _v, SZ, _v],
    'pbsed_allreduce_finish': [_v, _v],
}
_NON_STATUS = {'pbsed_gru_granule_capacity': C.c_int, 'pbsed_set_launch_cus': C.c_int, 'pbsed_gru_set_xcd_local': C.c_int, 'pbsed_conv_bwd_weight_bng_supported': C.c_int, 'pbsed_scratch_bytes': C.c_size_t, 'pbsed_last_error': C.c_char_p, 'pbsed_version': C.c_int, 'pbsed_comm_id_bytes': C.c_int, 'pbsed_conv_pack_dims': None,
               'pbsed_conv_pack_dims_bf16': None, 'pbsed_conv_pack_dims_wino': None, 'pbsed_conv_pack_dims_winox3': None, 'pbsed_conv_pack_dims_s16': None,
               'pbsed_conv1d_pack_dims_x3': None}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build the HIP extension first '
                f'(python -c "import __graft_entry__ as g; g.build()" or pb_sed_amd/csrc/build.sh). '
                f'pb_sed_amd has no CPU/eager fallback.')
        l = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if a declared symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _NON_STATUS.get(name, C.c_int)
        _lib = l
    return _lib


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), 'pb_sed_amd kernels need contiguous tensors'
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
    return arr


def int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


n_calls = 0            # C-ABI calls made so far (bench.py reports launches-class calls per step)
timing_filter = None   # optional predicate on the entry-point name: only those calls are bracketed
timing = None   # set to a list to record (name, tag, flops, bytes, start_event, end_event) per call


def call(name, *args, tag='', flops=0, nbytes=0):
    """Invoke a status-returning entry point; raise RuntimeError with the library's message.
    With ``timing`` enabled the call is bracketed by HIP events on the launch stream."""
    global n_calls
    n_calls += 1
    timed = timing is not None and (timing_filter is None or timing_filter(name))
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib().pbsed_last_error().decode()}')
    if timed:
        e1.record()
        timing.append((name, tag, flops, nbytes, e0, e1))


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('pb_sed_amd ops run on an MI355X (HIP) device only; got a CPU tensor. '
                           'There is no CPU fallback - use the oracle/ package for CPU checks.')
