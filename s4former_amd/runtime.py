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


# Residual stream (token tensors between the encoder layers and their gradients, vit.py:113-127): fp32 in parity mode; in
# bf16 mode it is kept in bf16 since round 3 (the proj / fc2 GEMM epilogues and both LayerNorm passes were HBM-bound on the
# fp32 residual traffic).  S4F_RESID=fp32 keeps it in fp32 in bf16 mode too (A/B switch, round-2 behaviour).
_resid_fp32 = os.environ.get('S4F_RESID', 'auto').lower() in ('fp32', 'f32', 'float32')


def set_residual_fp32(flag):
    global _resid_fp32
    _resid_fp32 = bool(flag)


def residual_dtype():
    """S4F_BF16 or S4F_F32: the `xdtype` argument of the C ABI"""
    return BF16 if (_code == BF16 and not _resid_fp32) else F32
