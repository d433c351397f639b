"""Process-wide numeric mode of the hot path.

'bf16' (default, perf mode): bf16 MFMA operands / activations, fp32 accumulate, fp32 master weights, residual
stream, statistics and losses.  'fp32' (parity mode): the same kernels instantiated on exact fp32 MFMA.
Select with set_compute_dtype() or the environment variable S4F_DTYPE=bf16|fp32."""
import os

from ._lib import BF16, F32

_NAMES = {'bf16': BF16, 'bfloat16': BF16, 'fp32': F32, 'f32': F32, 'float32': F32}
_code = _NAMES.get(os.environ.get('S4F_DTYPE', 'bf16').lower(), BF16)


def set_compute_dtype(name):
    global _code
    if name not in _NAMES:
        raise ValueError(f'unknown compute dtype {name!r}; use one of {sorted(_NAMES)}')
    _code = _NAMES[name]


def compute_dtype():
    """S4F_BF16 or S4F_F32 (the `dtype` argument of the C ABI)"""
    return _code


def compute_dtype_name():
    return 'bf16' if _code == BF16 else 'fp32'


# Residual stream (token tensors between the encoder layers and their gradients, vit.py:113-127): fp32 in both modes by
# default.  S4F_RESID=bf16 keeps it in bf16 in bf16 mode (kernels and tests exist: round 3).  Measured on the default workload
# (profiles/r03_*): the step gains 0.24 ms (30.76 -> 30.51 ms: the LayerNorm passes are latency-bound, not HBM-bound) while the
# gradient arena's cosine against the fp32 step falls from 0.99939 to 0.99872 at cfg2 (relative error 3.5 % -> 5.1 %) - not worth it.
_resid_fp32 = os.environ.get('S4F_RESID', 'fp32').lower() not in ('bf16', 'bfloat16')


def set_residual_fp32(flag):
    global _resid_fp32
    _resid_fp32 = bool(flag)


def residual_dtype():
    """S4F_BF16 or S4F_F32: the `xdtype` argument of the C ABI"""
    return BF16 if (_code == BF16 and not _resid_fp32) else F32


# Round 6 - the PRECISE TEACHER (S4F_TEACHER_PRECISE=1 / set_teacher_precise(True)): the teacher pass (backbone_ema + decode_head_ema:
# no gradients, 12.6 % of the step's flop) runs on the fp32 parity kernels (v_mfma_f32_16x16x4_f32 chains, fp32 activations) while the
# student keeps the bf16 perf mode.  The reference forms softmax / max / `> threshold` in fp32 (encoder_decoder.py:888-901); with bf16
# teacher activations 0.7 - 1.3 % of the pseudo-label entries sit on the other side of the 0.95 threshold (no argmax flips), with the
# fp32 teacher the pseudo-label masks and the PASA confidences are the parity mode's: equal to the reference's outside its tie set.
# The teacher's ParamStore simply takes dtype F32 (its kernels read store.dtype); the EMA writes no bf16 shadow for it.
_teacher_precise = os.environ.get('S4F_TEACHER_PRECISE', '0') not in ('0', '')


def set_teacher_precise(flag):
    global _teacher_precise
    _teacher_precise = bool(flag)


def teacher_precise():
    return _teacher_precise


def teacher_dtype():
    """numeric mode of the teacher's ParamStore: the compute dtype, or F32 under the precise-teacher switch"""
    return F32 if _teacher_precise else _code
