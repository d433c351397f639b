"""In-model strong augmentations of the S4Former "ours" configuration (configs/setr/..._MT_w_ours.py): CutMix between the
unlabeled student images + PatchShuffle, applied between the masked and the plain student pass
(reference mmseg/models/segmentors/encoder_decoder.py:633-638; mmseg/utils/generate_unsup_data.py:7-26,400-453,737-819).

Host side = the random DECISIONS only, drawn from numpy's and torch's global generators in exactly the reference's call
order (so that a seeded run reproduces the reference's augmentation); the pixels move in one HIP gather kernel
(kernels.mix_images / cutmix_labels), the decode head's token un-shuffle (decode_head.py:186-212) in kernels.gather_rows."""
import numpy as np
import torch


def _cutout_box(H, W, ratio):
    """generate_cutout_mask (generate_unsup_data.py:7-26) as the box it zeroes: (y0, y1, x0, x1)"""
    area = H * W / ratio
    w = np.random.randint(W / ratio + 1, W)
    h = np.round(area / w)
    x0 = np.random.randint(0, W - w + 1)
    y0 = np.random.randint(0, H - h + 1)
    return int(y0), int(y0 + h), int(x0), int(x0 + w)


def draw_strong_aug(B, H, W, strong_aug_prob, cutout_ratio, patchmix_ratio, block):
    """-> (boxes int32 [B, 4], perms int32 [B, G*G]).  Order of RNG calls as in the reference: one np.random.uniform (CutMix
    at all?), per image randint x3 (box), then per image np.random.rand (shuffle?) and torch.randperm over the blocks."""
    if not isinstance(cutout_ratio, int):
        raise ValueError('cutout_area must be an int (the tuple form draws from python\'s `random`, unused by the SETR configs)')
    boxes = np.zeros((B, 4), dtype=np.int32)
    if np.random.uniform(0, 1) < strong_aug_prob:
        for i in range(B):
            boxes[i] = _cutout_box(H, W, cutout_ratio)
    n = (H // block) * (W // block)
    perms = np.tile(np.arange(n, dtype=np.int32), (B, 1))
    for i in range(B):
        if np.random.rand() < patchmix_ratio:
            perms[i] = torch.randperm(n).numpy().astype(np.int32)
    return boxes, perms


def token_unshuffle_maps(perms, grid, n):
    """row maps over a [B, 1 + grid*grid, C] token tensor (cls first) for the un-shuffle of decode_head.py:186-212 with
    PatchMix_N = n: block q of the result is the block at the position p with perms[b][p] == q.
    Returns (fwd, bwd) int32 [B * (1 + grid*grid)]: unshuffled.flat_rows = tokens.flat_rows[fwd]; the adjoint (gradient
    rows back to their shuffled places) is the gather with bwd, the inverse permutation."""
    B = perms.shape[0]
    G = grid // n
    T = grid * grid
    rr, cc = np.meshgrid(np.arange(grid), np.arange(grid), indexing='ij')
    blk = (rr // n) * G + (cc // n)                            # block index of every token position
    fwd = np.empty((B, T + 1), dtype=np.int64)
    for b in range(B):
        inv = np.empty(G * G, dtype=np.int64)
        inv[perms[b]] = np.arange(G * G)                       # inv[q] = p  with perms[p] == q
        p = inv[blk]                                           # source block position of every destination token
        src = ((p // G) * n + rr % n) * grid + (p % G) * n + cc % n
        fwd[b, 0] = 0
        fwd[b, 1:] = 1 + src.reshape(-1)
    fwd += (np.arange(B, dtype=np.int64) * (T + 1))[:, None]
    fwd = fwd.reshape(-1)
    bwd = np.empty_like(fwd)
    bwd[fwd] = np.arange(fwd.size)
    return fwd.astype(np.int32), bwd.astype(np.int32)
