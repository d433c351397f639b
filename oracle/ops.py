"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in plain torch fp32, of the arithmetic of every op on the S4Former hot path.  Each function
cites the reference file:line it follows (paths relative to the reference repo JoyHuYY1412/S4Former).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Parity status: pinned (a) by the reference's own known-answer CE tests (the answers of reference
tests/test_models/test_losses/test_ce_loss.py, restated inline in tests/test_oracle_golden.py::test_reference_ce_known_answers
and tests/test_kernels_gpu.py::test_ce_known_answers) and (b) by golden vectors generated in the build
container from the reference's own hot-path files imported under a minimal mmcv stand-in
(tests/golden/make_golden.py -> tests/golden/*.npz).
"""
import math

import torch
import torch.nn.functional as F


def linear(x, w, b=None):
    """mmcv FFN / nn.MultiheadAttention projections (vit.py:86-103): y = x W^T + b."""
    return F.linear(x, w, b)


def gelu(x):
    """nn.GELU() exact erf form (mmcv FFN act_cfg=dict(type='GELU'), vit.py:86-95)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layernorm(x, gamma, beta, eps=1e-6):
    """nn.LayerNorm(768, eps=1e-6) (vit.py:67-69,82-84; setr_up_head.py:49,103): biased variance."""
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, eps)


def patch_embed(img, w, b):
    """PatchEmbed.forward (models/utils/embed.py:183-204): Conv2d k=s=16, flatten(2).transpose(1,2)."""
    y = F.conv2d(img, w, b, stride=16)
    hw = (y.shape[2], y.shape[3])
    return y.flatten(2).transpose(1, 2), hw


def assemble_tokens(patches, cls_token, pos_embed):
    """vit.py:486-487,445: x = cat(cls, patches) + pos_embed (dropout p = 0)."""
    B = patches.shape[0]
    return torch.cat((cls_token.expand(B, -1, -1), patches), dim=1) + pos_embed


def attention_core(qkv, num_heads, bias=None):
    """nn.MultiheadAttention core as used by mmcv MultiheadAttention (vit.py:99-103):
    q scaled by head_dim^-0.5 before the product, additive float mask [B, N, N] (same for all heads of an
    image, vit.py:534-535), softmax over keys, P v, heads concatenated.  qkv [B, N, 3*h*dh] -> ctx [B, N, h*dh]."""
    B, N, three_c = qkv.shape
    c = three_c // 3
    dh = c // num_heads
    q, k, v = qkv.split(c, dim=-1)
    q = q.reshape(B, N, num_heads, dh).transpose(1, 2) * (dh ** -0.5)
    k = k.reshape(B, N, num_heads, dh).transpose(1, 2)
    v = v.reshape(B, N, num_heads, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias[:, None, :, :]
    p = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    ctx = (p @ v).transpose(1, 2).reshape(B, N, c)
    return ctx, lse


def pasa_bias(u, weight, adaptive):
    """vit.py:519-535.  u [B, n_patches] = per-patch mean of (1 - conf).  Returns the additive mask [B, N, N]
    (N = n_patches + 1, cls first with u = 0): bias[b, i, j] = weight * u_j, rows of the more-confident half
    of the patch tokens zeroed when adaptive."""
    B = u.shape[0]
    um = torch.cat((torch.zeros(B, 1, dtype=u.dtype), u.reshape(B, -1)), dim=-1)
    N = um.shape[-1]
    m = um.unsqueeze(1).repeat(1, N, 1)
    if adaptive:
        idx = torch.topk(um[:, 1:], int(0.5 * (N - 1)), dim=-1, largest=False)[1] + 1
        m[torch.arange(B).unsqueeze(1), idx, :] = 0
    return m * weight


def pasa_rank1(u, adaptive):
    """The same mask in the rank-1 form the kernel takes: (bias_u [B,N], row_flag [B,N])."""
    B = u.shape[0]
    um = torch.cat((torch.zeros(B, 1, dtype=u.dtype), u.reshape(B, -1)), dim=-1)
    N = um.shape[-1]
    flag = torch.ones(B, N, dtype=u.dtype)
    if adaptive:
        idx = torch.topk(um[:, 1:], int(0.5 * (N - 1)), dim=-1, largest=False)[1] + 1
        flag[torch.arange(B).unsqueeze(1), idx] = 0
    return um, flag


def conv3x3(x, w):
    """ConvModule's Conv2d(k=3, pad=1, stride=1, bias=False) (setr_up_head.py:57-64). NCHW."""
    return F.conv2d(x, w, None, stride=1, padding=1)


def batchnorm_train(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    """(Sync)BatchNorm2d in train mode (configs/setr/*:11): batch statistics over (B,H,W), running stats updated
    in place (unbiased variance)."""
    return F.batch_norm(x, running_mean, running_var, gamma, beta, True, momentum, eps)


def batchnorm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5):
    return F.batch_norm(x, running_mean, running_var, gamma, beta, False, 0.0, eps)


def upsample(x, scale):
    """mmseg.ops.Upsample (ops/wrappers.py:46-51): size = int(in * scale); F.interpolate(bilinear,
    align_corners=False)."""
    size = [int(t * scale) for t in x.shape[-2:]]
    return F.interpolate(x, size, None, 'bilinear', False)


def ce_mean_all(logits, label, ignore_index=255, loss_weight=1.0, class_weight=None):
    """CrossEntropyLoss.forward with reduction='mean', avg_non_ignore=False (cross_entropy_loss.py:45-61,
    losses/utils.py:68-69): per-pixel CE with ignored pixels contributing 0, mean over ALL pixels."""
    loss = F.cross_entropy(logits, label, weight=class_weight, reduction='none', ignore_index=ignore_index)
    return loss_weight * loss.mean()


def ce_none(logits, label, ignore_index=-100, class_weight=None):
    return F.cross_entropy(logits, label, weight=class_weight, reduction='none', ignore_index=ignore_index)


def pseudo_label(seg_logits, th):
    """extract_teacher_info_ema + foward_unsup_train (encoder_decoder.py:888-901,541-542)."""
    max_value, label = torch.max(F.softmax(seg_logits, dim=1), dim=1)
    conf = (max_value > th) * 1
    label = label.clone()
    label[conf == 0] = 255
    return label, conf


def ema_update(tgt, src, momentum):
    """update_ema_variables (encoder_decoder.py:1058-1060): tgt.mul_(m).add_(src, alpha=1-m), in place."""
    tgt.mul_(momentum).add_(src, alpha=1 - momentum)
    return tgt


def poly_lr(base_lr, it, max_iters, power=0.9, min_lr=1e-4):
    """mmcv PolyLrUpdaterHook (schedule_80k_pascal_1over8.py:5): by_epoch=False."""
    coeff = (1 - it / max_iters) ** power
    return (base_lr - min_lr) * coeff + min_lr


# ---------------------------------------------------------------------------------------------- "ours" additions (§8f-1)
def cutout_box(img_size, ratio=2):
    """generate_cutout_mask (mmseg/utils/generate_unsup_data.py:7-26) as the box (y0, y1, x0, x1) it zeroes, drawing from
    numpy's GLOBAL generator in the reference's call order (randint w, randint x_start, randint y_start)."""
    import numpy as np
    H, W = img_size
    area = H * W / ratio
    w = np.random.randint(W / ratio + 1, W)
    h = np.round(area / w)
    x0 = np.random.randint(0, W - w + 1)
    y0 = np.random.randint(0, H - h + 1)
    return int(y0), int(y0 + h), int(x0), int(x0 + w)


def cutmix(img, label, boxes):
    """generate_unsup_cutmix_data (generate_unsup_data.py:400-453, patchwise=False): image / label i keeps itself where
    mask == 1 and takes sample (i + 1) % B inside box i.  (The reference routes the labels through a float + nearest
    resize round trip of identical size: the identity.)"""
    B = img.shape[0]
    new_img, new_lab = img.clone(), label.clone()
    for i, (y0, y1, x0, x1) in enumerate(boxes):
        j = (i + 1) % B
        new_img[i, :, y0:y1, x0:x1] = img[j, :, y0:y1, x0:x1]
        new_lab[i, y0:y1, x0:x1] = label[j, y0:y1, x0:x1]
    return new_img, new_lab


def patch_shuffle(img, perms, block):
    """generate_unsup_patchmix_data (generate_unsup_data.py:737-819): block position p of image b receives block perms[b][p]
    (row-major block index over the (H / block) x (W / block) grid)."""
    B, C, H, W = img.shape
    G = W // block
    out = img.clone()
    for b in range(B):
        for p in range(G * G):
            s = int(perms[b][p])
            out[b, :, (p // G) * block:(p // G + 1) * block, (p % G) * block:(p % G + 1) * block] = \
                img[b, :, (s // G) * block:(s // G + 1) * block, (s % G) * block:(s % G + 1) * block]
    return out


def draw_strong_aug(B, img_size, strong_aug_prob=0.5, cutout_ratio=2, patchmix_ratio=0.5, block=128):
    """the random decisions of the use_PatchShuffle_w_Cutmix branch (encoder_decoder.py:633-638) in the reference's order of
    RNG calls: np.random.uniform (CutMix at all?), per image the cut-out box, then per image np.random.rand (shuffle?) and
    torch.randperm over the blocks.  Returns (boxes [(y0, y1, x0, x1)] - empty boxes when CutMix is skipped, perms)."""
    import numpy as np
    H, W = img_size
    if np.random.uniform(0, 1) < strong_aug_prob:
        boxes = [cutout_box(img_size, cutout_ratio) for _ in range(B)]
    else:
        boxes = [(0, 0, 0, 0)] * B
    n = (H // block) * (W // block)
    perms = []
    for _ in range(B):
        if np.random.rand() < patchmix_ratio:
            perms.append(torch.arange(n)[torch.randperm(n)])
        else:
            perms.append(torch.arange(n))
    return boxes, perms


def repatchmix_tokens(tokens, perms, n):
    """BaseDecodeHead._repatchmix_inputs (decode_heads/decode_head.py:186-212): tokens [B, T, C] of a patch-shuffled image
    (T = g x g patches, shuffled in blocks of n x n patches) back in the original block order: block q of the result is the
    block at the position p with perms[b][p] == q."""
    B, T, C = tokens.shape
    g = int(round(T ** 0.5))
    G = g // n
    x = tokens.reshape(B, G, n, G, n, C)
    out = torch.empty_like(x)
    for b in range(B):
        for p in range(G * G):
            q = int(perms[b][p])
            out[b, q // G, :, q % G] = x[b, p // G, :, p % G]
    return out.reshape(B, T, C)


def ncr_unsup_only(student_logits, teacher_logits, hard_label):
    """compute_pseudo_loss, negative_class_ranking_mode == 'unsup_only' (encoder_decoder.py:936-954): per class c the pixels
    labelled c, student / teacher logits without channel c, softmax over the rest, nn.PairwiseDistance(p=2) (eps 1e-6 added
    to the difference), summed, divided by B H W."""
    B, C, H, W = teacher_logits.shape
    s = student_logits.permute(0, 2, 3, 1)
    t = teacher_logits.permute(0, 2, 3, 1)
    pdist = torch.nn.PairwiseDistance(p=2)
    loss = 0
    for c in range(C):
        m = hard_label == c
        sc = torch.cat((s[m][:, :c], s[m][:, c + 1:]), dim=1)
        tc = torch.cat((t[m][:, :c], t[m][:, c + 1:]), dim=1)
        loss = loss + torch.sum(pdist(F.softmax(sc, dim=1), F.softmax(tc, dim=1)))
    return loss / (B * H * W)


# ---------------------------------------------------------------------------------------------- evaluation path (§8f-2)
def resize(x, size, align_corners=False):
    """mmseg.ops.resize (ops/wrappers.py:8-51) with mode='bilinear'"""
    return F.interpolate(x, tuple(int(s) for s in size), None, 'bilinear', align_corners)


def whole_inference_post(seg_logit, img_shape, ori_shape, flip=None, align_corners=False):
    """whole_inference + inference + simple_test after the network (encoder_decoder.py:1127-1147, 1193-1216): remove the padding
    area, rescale to ori_shape, softmax, flip back, arg-max.  -> (prob [B, C, H, W], label int64 [B, H, W])"""
    seg_logit = seg_logit[:, :, :img_shape[0], :img_shape[1]]
    seg_logit = resize(seg_logit, ori_shape[:2], align_corners)
    out = F.softmax(seg_logit, dim=1)
    if flip == 'horizontal':
        out = out.flip(dims=(3,))
    elif flip == 'vertical':
        out = out.flip(dims=(2,))
    return out, out.argmax(dim=1)


def intersect_and_union(pred_label, label, num_classes, ignore_index, label_map=None, reduce_zero_label=False):
    """core/evaluation/metrics.py:26-85 (array inputs): three torch.histc passes over the non-ignored pixels"""
    pred_label = torch.as_tensor(pred_label).clone()
    label = torch.as_tensor(label).clone()
    if label_map:
        src = label.clone()
        for old_id, new_id in label_map.items():
            label[src == old_id] = new_id
    if reduce_zero_label:
        label[label == 0] = 255
        label = label - 1
        label[label == 254] = 255
    mask = label != ignore_index
    pred_label, label = pred_label[mask], label[mask]
    intersect = pred_label[pred_label == label]
    ai = torch.histc(intersect.float(), bins=num_classes, min=0, max=num_classes - 1)
    ap = torch.histc(pred_label.float(), bins=num_classes, min=0, max=num_classes - 1)
    al = torch.histc(label.float(), bins=num_classes, min=0, max=num_classes - 1)
    return ai, ap + al - ai, ap, al


def mean_iou(results, gt_seg_maps, num_classes, ignore_index, reduce_zero_label=False):
    """metrics.py:88-166, 330-365: totals in float64, aAcc / IoU / Acc"""
    tot = [torch.zeros(num_classes, dtype=torch.float64) for _ in range(4)]
    for r, g in zip(results, gt_seg_maps):
        for t, a in zip(tot, intersect_and_union(r, g, num_classes, ignore_index, None, reduce_zero_label)):
            t += a
    ai, au, ap, al = tot
    return dict(aAcc=(ai.sum() / al.sum()).numpy(), IoU=(ai / au).numpy(), Acc=(ai / al).numpy()), tot


# ---------------------------------------------------------------------------------------------- input pipeline (§8f-3)
# numpy restatement of the per-view pipeline after Resize (mmseg/datasets/pipelines/transforms.py:429-611, 802-875, 1165-1285).
# cv2 is NOT in the build image: bgr2hsv / hsv2bgr restate OpenCV's published 8-bit algorithm (RGB2HSV_b with the 12-bit
# division tables, HSV2RGB_b through the float sector formula) - PARITY UNPINNED for these two functions; everything else is
# numpy arithmetic exactly as the reference writes it.
def _bgr2hsv_u8(img):
    import numpy as np
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    v = np.maximum(b, np.maximum(g, r))
    vmin = np.minimum(b, np.minimum(g, r))
    diff = v - vmin
    with np.errstate(divide='ignore'):
        sdiv = np.where(v > 0, np.rint((255 << 12) / (1.0 * np.maximum(v, 1))), 0).astype(np.int64)
        hdiv = np.where(diff > 0, np.rint((180 << 12) / (6.0 * np.maximum(diff, 1))), 0).astype(np.int64)
    s = (diff * sdiv + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv + (1 << 11)) >> 12
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], -1).astype(np.uint8)


def _hsv2bgr_u8(hsv):
    import numpy as np
    f32 = np.float32
    h = hsv[..., 0].astype(f32) * f32(6.0 / 180.0)
    s = hsv[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = hsv[..., 2].astype(f32) * f32(1.0 / 255.0)
    sector = np.floor(h).astype(np.int64)
    fr = (h - sector.astype(f32)).astype(f32)
    sector = np.where((sector < 0) | (sector >= 6), 0, sector)
    t0, t1 = v, (v * (f32(1) - s)).astype(f32)
    t2 = (v * (f32(1) - (s * fr).astype(f32))).astype(f32)
    t3 = (v * (f32(1) - (s * (f32(1) - fr)).astype(f32))).astype(f32)
    tab = np.stack([t0, t1, t2, t3], -1)
    sel = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])[sector]       # (b, g, r) table index
    bgr = np.take_along_axis(tab, sel, -1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr)
    return np.clip(np.rint((bgr * f32(255)).astype(f32)), 0, 255).astype(np.uint8)


def photometric(img, p):
    """PhotoMetricDistortion with the drawn parameters p = (bright on, delta, contrast on, alpha, contrast first, sat on, alpha,
    hue on, delta) on a uint8 BGR image (transforms.py:1197-1268)"""
    import numpy as np

    def convert(x, alpha=1, beta=0):
        return np.clip(x.astype(np.float32) * np.float32(alpha) + np.float32(beta), 0, 255).astype(np.uint8)
    if p[0]:
        img = convert(img, beta=p[1])
    if p[2] and p[4]:
        img = convert(img, alpha=p[3])
    if p[5]:
        hsv = _bgr2hsv_u8(img)
        hsv[..., 1] = convert(hsv[..., 1], alpha=p[6])
        img = _hsv2bgr_u8(hsv)
    if p[7]:
        hsv = _bgr2hsv_u8(img)
        hsv[..., 0] = (hsv[..., 0].astype(int) + int(p[8])) % 180
        img = _hsv2bgr_u8(hsv)
    if p[2] and not p[4]:
        img = convert(img, alpha=p[3])
    return img


def rescale_size(old_wh, scale):
    """mmcv.image.geometric.rescale_size (mmcv-full 1.4.4 - 1.6.0, absent here: restated): (w, h), scale = factor or
    (edge, edge) -> (new_w, new_h) = int(x * factor + 0.5), factor = min(long / max(h, w), short / min(h, w))"""
    w, h = old_wh
    if isinstance(scale, (float, int)):
        factor = scale
    else:
        factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


def cv_resize_linear_u8(img, new_hw):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_LINEAR) on uint8 [H, W, C] (what mmcv.imrescale calls for the
    image, transforms.py:370).  cv2 is absent: OpenCV's published 8-bit algorithm restated (imgproc/resize.cpp: coordinates
    (d + 0.5) * scale - 0.5 in float, weights rounded to 1 / 2048 as shorts, horizontal pass in int32, vertical pass
    (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2; an exact factor-2 shrink is the 2 x 2 area mean).
    PARITY UNPINNED: kernel and this restatement agree bit for bit, neither has met cv2."""
    import numpy as np
    H, W = img.shape[:2]
    RH, RW = int(new_hw[0]), int(new_hw[1])
    if (RH, RW) == (H, W):
        return img.copy()
    if RW * 2 == W and RH * 2 == H:
        a = img.astype(np.int32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    scale_x, scale_y = 1.0 / (RW / W), 1.0 / (RH / H)

    def axis(n_dst, n_src, scale, clamp_weight):
        f = ((np.arange(n_dst) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = (f - s0.astype(np.float32)).astype(np.float32)
        if clamp_weight:                              # horizontal: fx = 0 where the pair would leave the row
            lo, hi = s0 < 0, s0 >= n_src - 1
            f = np.where(lo | hi, np.float32(0), f)
            s0 = np.where(lo, 0, np.where(hi, n_src - 1, s0))
        w0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int32)
        w1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int32)
        i0 = np.clip(s0, 0, n_src - 1)
        i1 = np.clip(s0 + 1, 0, n_src - 1)
        return i0, i1, w0, w1
    x0, x1, a0, a1 = axis(RW, W, scale_x, True)
    y0, y1, b0, b1 = axis(RH, H, scale_y, False)
    src = img.astype(np.int32)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]            # [H, RW, C]
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def cv_resize_nearest(seg, new_hw):
    """cv2.resize(seg, (new_w, new_h), interpolation=cv2.INTER_NEAREST) (mmcv.imrescale(..., 'nearest'), transforms.py:393):
    source index min(floor(d * scale), size - 1) with scale = 1. / (dst / src) in double.  PARITY UNPINNED (cv2 absent)."""
    import numpy as np
    H, W = seg.shape[:2]
    RH, RW = int(new_hw[0]), int(new_hw[1])
    sy = np.minimum(np.floor(np.arange(RH) * (1.0 / (RH / H))).astype(np.int64), H - 1)
    sx = np.minimum(np.floor(np.arange(RW) * (1.0 / (RW / W))).astype(np.int64), W - 1)
    return seg[sy][:, sx]


def input_view(img, seg, bbox, flip, p, crop_size, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True,
               resize_to=None, flip_direction='horizontal'):
    """[Resize ->] RandomCrop.crop -> RandomFlip -> PhotoMetricDistortion -> Normalize -> Pad -> CHW
    (transforms.py:171-427, 826-832, 450-481, 1241-1268, 589-611, 541-569).  -> (img fp32 [3, ch, cw], seg uint8 [ch, cw])"""
    import numpy as np
    if resize_to is not None:
        img = cv_resize_linear_u8(img, resize_to)
        seg = cv_resize_nearest(seg, resize_to) if seg is not None else None
    y1, y2, x1, x2 = bbox
    img = img[y1:y2, x1:x2]
    seg = seg[y1:y2, x1:x2] if seg is not None else None
    if flip:
        if flip_direction == 'horizontal':
            img = img[:, ::-1]
            seg = seg[:, ::-1] if seg is not None else None
        else:
            img = img[::-1]
            seg = seg[::-1] if seg is not None else None
    img = photometric(np.ascontiguousarray(img), p).astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    stdinv = (1.0 / np.asarray(std, dtype=np.float64)).astype(np.float32)
    img = (img - np.asarray(mean, dtype=np.float32)) * stdinv
    out = np.zeros((crop_size[0], crop_size[1], 3), dtype=np.float32)
    out[:img.shape[0], :img.shape[1]] = img
    so = None
    if seg is not None:
        so = np.full(crop_size, 255, dtype=np.uint8)
        so[:seg.shape[0], :seg.shape[1]] = seg
    return np.ascontiguousarray(out.transpose(2, 0, 1)), so
