"""ctypes binding of libs4f_hip.so (the C ABI declared in include/s4f.h).

The product path has no CPU fallback: if the library is missing this module raises at first use.
Every wrapper takes torch tensors (device memory is owned by the caller = torch), passes raw pointers and the
current HIP stream, and raises S4FError with the library's message on a non-zero return.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

import torch

F32, BF16 = 0, 1
OP_ROW, OP_K, OP_ROW_CONV, OP_K_TAPSPLIT, OP_K_CONV = 0, 1, 2, 3, 4
ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_COLSTATS = 0, 1, 2, 3

_LIB_PATH = os.environ.get('S4F_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libs4f_hip.so')   # S4F_LIB: A/B builds
_lib = None


class S4FError(RuntimeError):
    pass


class GemmDesc(Structure):
    _fields_ = [
        ('A', c_void_p), ('B', c_void_p),
        ('M', c_int32), ('N', c_int32), ('K', c_int32),
        ('lda', c_int64), ('ldb', c_int64),
        ('a_mode', c_int32), ('b_mode', c_int32),
        ('dtype', c_int32), ('splitk', c_int32),
        ('cB', c_int32), ('cH', c_int32), ('cW', c_int32), ('cC', c_int32), ('csign', c_int32),
        ('alpha', c_float),
        ('bias', c_void_p), ('resid', c_void_p), ('ldr', c_int64),
        ('out_f32', c_void_p), ('ldo_f32', c_int64),
        ('out_t', c_void_p), ('ldo_t', c_int64),
        ('out_pre', c_void_p), ('ldo_pre', c_int64),
        ('aux', c_void_p), ('ld_aux', c_int64),
        ('act', c_int32), ('atomic', c_int32),
        ('pos_period', c_int32), ('tile_hint', c_int32), ('pos', c_void_p),
        ('colsum', c_void_p),
        ('resid_t', c_int32), ('gelu_q8', c_int32),
    ]
assert ctypes.sizeof(GemmDesc) == 216, 'GemmDesc must mirror s4f_gemm_desc (include/s4f.h, static_assert in gemm.hip)'


class LayerDesc(Structure):
    """mirror of s4f_layer_desc (include/s4f.h)"""
    _fields_ = ([(n, c_int32) for n in ('B', 'N', 'E', 'F', 'H', 'dtype', 'xdtype')] + [('eps', c_float), ('bias_w', c_float)] +
                [('hint', c_int32 * 8), ('wg_hint', c_int32), ('wg_splitk', c_int32), ('fold_colsum', c_int32), ('gelu_q8', c_int32)] +
                [(n, c_void_p) for n in (
                    'ln1_g', 'ln1_b', 'ln2_g', 'ln2_b', 'bqkv', 'bo', 'b1', 'b2', 'wqkv', 'wo', 'w1', 'w2', 'wqkv_T', 'wo_T', 'w1_T', 'w2_T',
                    'bias_u', 'row_flag',
                    'x', 'xn', 'mean1', 'rstd1', 'qkv', 'ctx', 'lse', 'x1', 'xn2', 'mean2', 'rstd2', 'gelu_d', 'a', 'x2',
                    'g2', 'g2t', 'g2cs', 'dz', 'dxn2', 'g1', 'g1t', 'dctx', 'dqkv', 'delta', 'dxn', 'g0', 'g0t', 'g0cs',
                    'd_ln1_g', 'd_ln1_b', 'd_ln2_g', 'd_ln2_b', 'd_wqkv', 'd_bqkv', 'd_wo', 'd_bo', 'd_w1', 'd_b1', 'd_w2', 'd_b2', 'attn_ws')] +
                [('attn_ws_bytes', c_int64)])


assert ctypes.sizeof(LayerDesc) == 568, 'LayerDesc must mirror s4f_layer_desc (include/s4f.h, static_assert in layer.hip)'


_SIGS = {
    's4f_encoder_layer_fwd': [POINTER(LayerDesc), c_void_p],
    's4f_encoder_layer_bwd': [POINTER(LayerDesc), c_void_p, c_void_p, c_void_p],
    's4f_gemm': [POINTER(GemmDesc), c_void_p],
    's4f_gemm_grouped': [POINTER(GemmDesc), c_int, c_void_p],
    's4f_cast': [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    's4f_cast_back': [c_void_p, c_void_p, c_int64, c_int, c_void_p],
    's4f_transpose_many': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    's4f_im2col_patch16': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_cls_pos': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    's4f_tokens_bwd': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    's4f_colsum': [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    's4f_layernorm_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64,
                          c_float, c_int, c_int, c_void_p],
    's4f_layernorm_bwd': [c_void_p] * 11 + [c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_void_p],
    's4f_add_f32': [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    's4f_attention_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int,
                          c_void_p],
    's4f_attention_bwd': [c_void_p] * 8 + [c_float, c_int, c_int, c_int, c_int, c_void_p],
    's4f_attention_bwd_fused': [c_void_p] * 8 + [c_float, c_int, c_int, c_int, c_void_p, c_int64, c_void_p],
    's4f_attention_bwd_ws_bytes': [c_int, c_int, c_int],
    's4f_workspace_bytes': [c_int, c_void_p, c_int],
    's4f_bn_stats': [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p],
    's4f_bn_finalize': [c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int,
                        c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    's4f_bn_relu_up_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_bn_relu_up_bwd': [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_bn_bwd_apply': [c_void_p] * 6 + [c_double, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p],
    's4f_bn_param_grads': [c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    's4f_bn_relu_cls_fwd': [c_void_p] * 6 + [c_int, c_void_p, c_int64, c_int, c_int, c_int, c_void_p],
    's4f_cls_bn_bwd_stats': [c_void_p, c_int] + [c_void_p] * 9 + [c_int64, c_int, c_int, c_int, c_void_p],
    's4f_cls_bn_bwd_apply': [c_void_p, c_int] + [c_void_p] * 8 + [c_double, c_void_p, c_int64, c_int, c_int, c_int, c_void_p],
    's4f_upce_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_upce_bwd': [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                     c_int, c_int, c_int, c_int, c_void_p],
    's4f_up_pseudo_label': [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_void_p],
    's4f_up_logits_nchw': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_ncr_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_ncr_bwd': [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                    c_void_p],
    's4f_mix_images': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_cutmix_labels': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    's4f_gather_rows': [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p],
    's4f_pasa_patch_u': [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_resize_bilinear_nchw': [c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_int, c_int, c_int, c_void_p],
    's4f_softmax_argmax_nchw': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    's4f_confusion_counts': [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p],
    's4f_input_view': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int), c_int, POINTER(c_float),
                       POINTER(c_float), POINTER(c_float), c_int, c_float, c_int, c_void_p],
    's4f_input_view_resized': [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), c_int,
                               POINTER(c_float), POINTER(c_float), POINTER(c_float), c_int, c_float, c_int, c_void_p],
    's4f_ce_fwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p],
    's4f_ce_bwd': [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p],
    's4f_ema': [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_int, c_void_p],
    's4f_ema_to': [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_int, c_void_p],
    's4f_sgd_momentum': [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_int, c_int,
                         c_void_p],
}
_RESTYPES = {'s4f_attention_bwd_ws_bytes': c_int64, 's4f_workspace_bytes': c_int64}     # everything else returns the int status
EXPORTED_SYMBOLS = sorted(list(_SIGS) + ['s4f_last_error', 's4f_version'])


def lib_path():
    return _LIB_PATH


def load():
    """Load libs4f_hip.so (once). Raises S4FError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise S4FError(f'{_LIB_PATH} is missing: build it with `python -m s4former_amd.build` '
                       '(there is no CPU fallback for the product path)')
    lib = ctypes.CDLL(_LIB_PATH)
    lib.s4f_last_error.restype = c_char_p
    lib.s4f_last_error.argtypes = []
    lib.s4f_version.restype = c_int
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = _RESTYPES.get(name, c_int)
        fn.argtypes = sig
    _lib = lib
    return lib


_prof = None


class CallProfiler:
    """Times every C-ABI launch with HIP events recorded on the launch stream (markers only: nothing is
    serialised).  Used by bench.py for the live per-kernel durations behind the `roofline` object."""

    def __init__(self):
        self.entries = []
        self.entries_raw = []
        self.base = None

    def __enter__(self):
        global _prof
        self._prev, _prof = _prof, self
        self.base = torch.cuda.Event(enable_timing=True)
        self.base.record()
        return self

    def timeline(self):
        """[(name, tag, raw stream, start ms after __enter__, duration ms)] in issue order, after a device synchronise: the
        GPU-side schedule of the profiled region WITHOUT a tracer attached (tools/event_timeline.py)"""
        torch.cuda.synchronize()
        return [(name, tag, st, self.base.elapsed_time(e0), e0.elapsed_time(e1)) for name, tag, e0, e1, st in self.entries_raw]

    def __exit__(self, *exc):
        global _prof
        _prof = self._prev
        return False

    def summary(self):
        """{(name, tag): dict(calls, ms)} after a device synchronise"""
        torch.cuda.synchronize()
        out = {}
        for name, tag, e0, e1 in self.entries:   # (entries_raw carries the stream as a fifth field)
            d = out.setdefault((name, tag), dict(calls=0, ms=0.0))
            d['calls'] += 1
            d['ms'] += e0.elapsed_time(e1)
        return out


def call(name, *args, tag=None):
    lib = load()
    if _prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        _prof.entries.append((name, tag, e0, e1))
        _prof.entries_raw.append((name, tag, e0, e1, stream()))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise S4FError(f'{name} failed ({rc}): {lib.s4f_last_error().decode(errors="replace")}')


def dt(dtype):
    if dtype in (BF16, torch.bfloat16):
        return BF16
    if dtype in (F32, torch.float32):
        return F32
    raise S4FError(f'unsupported compute dtype {dtype}')


def torch_dtype(code):
    return torch.bfloat16 if code == BF16 else torch.float32


def p(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise S4FError('s4f kernels need device tensors (got a CPU tensor); there is no CPU fallback')
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def stream():
    """raw hipStream_t of torch's current stream (the C entry points: torch.cuda.current_stream() costs ~9 us per call
    on the host, this ~0.3 us; the step makes ~700 launches)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream
