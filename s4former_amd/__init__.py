"""s4former_amd — MI355X-native S4Former (DeiT-B / SETR-PUP mean-teacher) training step.

Host side: the reference's registry/plugin API (EncoderDecoder / VisionTransformer / SETRUPHead /
CrossEntropyLoss, configs load unchanged).  Device side: libs4f_hip.so, hand-written gfx950 HIP kernels behind
the C ABI of include/s4f.h.  There is no CPU fallback on the product path."""
from . import runtime  # noqa: F401
from ._lib import S4FError  # noqa: F401
from .checkpoint import convert_mmcls_deit, load_checkpoint, resume, save_checkpoint  # noqa: F401
from .config import Config  # noqa: F401
from .registry import (BACKBONES, HEADS, LOSSES, MODELS, SEGMENTORS, build_backbone, build_head, build_loss,  # noqa: F401
                       build_segmentor)
from .runtime import compute_dtype_name, set_compute_dtype  # noqa: F401


def _register_all():
    from . import encoder_decoder, losses, setr_up_head, vit  # noqa: F401


_register_all()
from .encoder_decoder import EncoderDecoder  # noqa: E402,F401
from .losses import CrossEntropyLoss  # noqa: E402,F401
from .optim import PolyLR, S4FSGD, build_optimizer  # noqa: E402,F401
from .setr_up_head import SETRUPHead  # noqa: E402,F401
from .vit import VisionTransformer  # noqa: E402,F401
