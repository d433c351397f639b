"""Repo-owned model dicts with the values the reference's configs resolve to (configs/setr/*_MT.py:137-241 and
configs/_base_/models/setr_pup.py), a deterministic synthetic batch generator and the algorithmic
FLOP accounting of BASELINE.md §3."""
import copy

import torch

NORM_BB = dict(type='LN', eps=1e-6, requires_grad=True)
NORM_HEAD = dict(type='SyncBN', requires_grad=True)


def setr_pup_model(img=512, embed=768, layers=12, heads=12, channels=256, num_classes=21, out_indices=(4, 7, 9, 11), **flags):
    backbone = dict(type='VisionTransformer', img_size=(img, img), patch_size=16, in_channels=3, embed_dims=embed,
                    num_layers=layers, num_heads=heads, out_indices=tuple(out_indices), drop_rate=0.0, norm_cfg=dict(NORM_BB),
                    with_cls_token=True, interpolate_mode='bilinear')
    decode = dict(type='SETRUPHead', in_channels=embed, channels=channels, in_index=3, num_classes=num_classes,
                  dropout_ratio=0, norm_cfg=dict(NORM_HEAD), num_convs=4, up_scale=2, kernel_size=3, align_corners=False,
                  loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    aux = [dict(type='SETRUPHead', in_channels=embed, channels=channels, in_index=i, num_classes=num_classes,
                dropout_ratio=0, norm_cfg=dict(NORM_HEAD), num_convs=2, up_scale=4, kernel_size=3, align_corners=False,
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)) for i in range(4)]
    cfg = dict(type='EncoderDecoder', pretrained=None, backbone=backbone, backbone_ema=copy.deepcopy(backbone),
               auxiliary_head=aux, decode_head=decode, decode_head_ema=copy.deepcopy(decode), ema=True, ema_momentum=0.999,
               unsup_weight=1.0, unsup_confidence=0.95, test_cfg=dict(mode='whole'))
    cfg.update(flags)
    return cfg


OPTIMIZER = dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0,
                 paramwise_cfg=dict(custom_keys={'head': dict(lr_mult=10.)}))
MAX_ITERS = 80001


def synthetic_batch(seed, n_sup, n_unsup, img=512, num_classes=21, block=32, border=8, device='cpu'):
    """img ~ N(0,1) clipped to [-2.2, 2.7]; labels = block x block squares of a uniform class with a
    border band of 255; the two views of an unlabeled image = same crop + independent N(0, 0.1^2) noise; tags in
    the order sup..., unsup_student..., unsup_teacher..., filenames matching student <-> teacher."""
    g = torch.Generator().manual_seed(seed)
    n = n_sup + 2 * n_unsup
    base = torch.randn(n_sup + n_unsup, 3, img, img, generator=g).clamp_(-2.2, 2.7)
    imgs = [base[:n_sup]]
    if n_unsup:
        u = base[n_sup:]
        imgs.append((u + 0.1 * torch.randn(u.shape, generator=g)).clamp_(-2.2, 2.7))
        imgs.append((u + 0.1 * torch.randn(u.shape, generator=g)).clamp_(-2.2, 2.7))
    imgs = torch.cat(imgs, 0).contiguous()
    nb = img // block
    cls = torch.randint(0, num_classes, (n, nb, nb), generator=g)
    gt = cls.repeat_interleave(block, 1).repeat_interleave(block, 2)
    band = torch.zeros(img, dtype=torch.bool)
    for s in range(0, img, block * 4):
        band[s:s + border] = True
    gt[:, band, :] = 255
    gt[:, :, band] = 255
    gt = gt.unsqueeze(1).contiguous()
    metas = [dict(tag='sup', filename=f'sup_{i}.jpg') for i in range(n_sup)]
    metas += [dict(tag='unsup_student', filename=f'{i}.jpg') for i in range(n_unsup)]
    metas += [dict(tag='unsup_teacher', filename=f'{i}.jpg') for i in range(n_unsup)]
    return imgs.to(device), gt.to(device), metas


def algorithmic_gflop(img=512, embed=768, layers=12, channels=256, num_classes=21):
    """forward GFLOP per image (2*MAC, literal reference graph)"""
    g = img // 16
    n = g * g + 1
    bb = layers * (12 * n * embed * embed + 2 * n * n * embed) * 2 + 2 * g * g * 768 * embed
    # 12 N d^2 MACs per layer = qkv (3) + proj (1) + ffn (8); attention 2 N^2 d MACs
    def head(num_convs, s):
        f, h, cin = 0, g, embed
        for _ in range(num_convs):
            f += 2 * h * h * 9 * cin * channels
            h, cin = h * s, channels
        f += 2 * h * h * channels * num_classes
        return f
    return dict(backbone=bb / 1e9, decode=head(4, 2) / 1e9, aux=head(2, 4) / 1e9)


def step_gflop(n_sup, n_unsup, img=512, num_classes=21, pseudo_loss=True, student_passes=1):
    """student_passes: 2 with attn_mask_seperate_head (the unlabeled images go through backbone + decode head masked AND plain)"""
    f = algorithmic_gflop(img=img, num_classes=num_classes)
    sup = 3 * (f['backbone'] + f['decode'] + 4 * f['aux'])
    uns = student_passes * 3 * (f['backbone'] + f['decode']) if pseudo_loss else f['backbone']
    tea = f['backbone'] + f['decode']
    return n_sup * sup + n_unsup * (uns + tea)
