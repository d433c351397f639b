"""ORACLE (test infrastructure, never shipped, never on the product path).

Plain-torch CPU restatement of the S4Former training step: DeiT/ViT backbone, SETR-PUP heads, CE with
mean-over-all-pixels, mean-teacher EMA, teacher pseudo-labels with confidence threshold, pseudo-label CE, PASA
attention bias, SGD + poly LR.  It keeps the reference's state-dict keys so that weights move between the
product model, this oracle and the reference with load_state_dict.  Every block cites the reference file:line
it follows (paths relative to JoyHuYY1412/S4Former).

Validated in the build container against the reference's own code (tests/golden/make_golden.py imports the
reference's hot-path files under a minimal mmcv stand-in and compares); golden vectors from that run are
committed under tests/golden/.
"""
import copy
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as O


class OracleLayer(nn.Module):
    """vit.py:28-127 TransformerEncoderLayer with mmcv MultiheadAttention(batch_first=True) and FFN."""

    def __init__(self, embed_dims, num_heads, ffn_channels, eps):
        super().__init__()
        self.ln1 = nn.LayerNorm(embed_dims, eps=eps)
        self.attn = nn.Module()
        self.attn.attn = nn.MultiheadAttention(embed_dims, num_heads, 0.0, bias=True)
        self.ln2 = nn.LayerNorm(embed_dims, eps=eps)
        self.ffn = nn.Module()
        self.ffn.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, ffn_channels), nn.GELU(), nn.Dropout(0.0)),
                                        nn.Linear(ffn_channels, embed_dims), nn.Dropout(0.0))

    def forward(self, x, attn_mask=None):
        # mmcv MultiheadAttention.forward: identity + proj_drop(attn(q, k, v, attn_mask)[0]), batch_first transposes
        q = self.ln1(x).transpose(0, 1)
        out = self.attn.attn(query=q, key=q, value=q, attn_mask=attn_mask)[0].transpose(0, 1)
        x = x + out
        # mmcv FFN.forward: identity + layers(x)
        return x + self.ffn.layers(self.ln2(x))


class OracleViT(nn.Module):
    """vit.py:129-577"""

    def __init__(self, img_size=(512, 512), patch_size=16, in_channels=3, embed_dims=768, num_layers=12, num_heads=12,
                 mlp_ratio=4, out_indices=(4, 7, 9, 11), eps=1e-6, **_):
        super().__init__()
        self.patch_size, self.num_heads, self.out_indices = patch_size, num_heads, list(out_indices)
        self.patch_embed = nn.Module()
        self.patch_embed.projection = nn.Conv2d(in_channels, embed_dims, patch_size, patch_size)
        n = (img_size[0] // patch_size) * (img_size[1] // patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dims))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dims))
        self.layers = nn.ModuleList([OracleLayer(embed_dims, num_heads, mlp_ratio * embed_dims, eps) for _ in range(num_layers)])

    def forward(self, inputs, attn_mask=None, attn_mask_weight=0.0, adaptive_attn_mask=False):
        B = inputs.shape[0]
        x, hw = O.patch_embed(inputs, self.patch_embed.projection.weight, self.patch_embed.projection.bias)
        x = O.assemble_tokens(x, self.cls_token, self.pos_embed)
        mask = None
        if attn_mask is not None:
            # vit.py:519-535
            m = O.pasa_bias(attn_mask.reshape(B, -1), attn_mask_weight, adaptive_attn_mask)
            mask = m.unsqueeze(1).repeat(1, self.num_heads, 1, 1).reshape(-1, m.size(-1), m.size(-1))
        outs = []
        for i, layer in enumerate(self.layers):
            x = layer(x, mask)
            if i in self.out_indices:
                out = x[:, 1:]
                outs.append(out.reshape(B, hw[0], hw[1], out.shape[-1]).permute(0, 3, 1, 2).contiguous())
        return tuple(outs)


class OracleConvModule(nn.Module):
    def __init__(self, cin, cout, eps=1e-5, momentum=0.1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=eps, momentum=momentum)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


class OracleUpsample(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = scale

    def forward(self, x):
        return O.upsample(x, self.scale)


class OracleHead(nn.Module):
    """setr_up_head.py:28-111 + decode_head.py:318-355"""

    def __init__(self, in_channels=768, channels=256, num_classes=21, num_convs=4, up_scale=2, in_index=3, loss_weight=1.0,
                 ln_eps=1e-6, ignore_index=255, **_):
        super().__init__()
        self.in_index, self.loss_weight, self.ignore_index, self.num_classes = in_index, loss_weight, ignore_index, num_classes
        self.conv_seg = nn.Conv2d(channels, num_classes, 1)
        self.norm = nn.LayerNorm(in_channels, eps=ln_eps)
        self.up_convs = nn.ModuleList()
        cin = in_channels
        for _ in range(num_convs):
            self.up_convs.append(nn.Sequential(OracleConvModule(cin, channels), OracleUpsample(up_scale)))
            cin = channels

    def forward(self, inputs, patchmix_n=0, perms=None):
        x = inputs[self.in_index]
        n, c, h, w = x.shape
        x = x.reshape(n, c, h * w).transpose(2, 1).contiguous()
        if patchmix_n:
            x = O.repatchmix_tokens(x, perms, patchmix_n)        # setr_up_head.py:98-100
        x = self.norm(x)
        x = x.transpose(1, 2).reshape(n, c, h, w).contiguous()
        for up in self.up_convs:
            x = up(x)
        return self.conv_seg(x)

    def losses(self, seg_logit, seg_label):
        seg_logit = F.interpolate(seg_logit, seg_label.shape[2:], None, 'bilinear', False)    # identity resize
        return {'loss_ce': O.ce_mean_all(seg_logit, seg_label.squeeze(1), self.ignore_index, self.loss_weight)}

    def forward_train(self, inputs, gt):
        return self.losses(self.forward(inputs), gt)


class OracleSegmentor(nn.Module):
    """encoder_decoder.py:25-163,386-687,875-934,1044-1066; base.py:230-274"""

    def __init__(self, backbone, decode_head, auxiliary_head=None, ema=True, ema_momentum=0.999, unsup_weight=1.0,
                 unsup_confidence=0.95, attn_mask_seperate_head=False, attn_mask_weight=50, adaptive_attn_mask=False,
                 fdrop_loss_weight=0.5, patchsize=16, plain_mt_pseudo_loss=False, use_PatchShuffle_w_Cutmix=False, PatchMix_N=8,
                 patchmix_ratio=0.5, strong_aug_prob=0.5, cutout_area=2, negative_class_ranking=False,
                 negative_class_ranking_mode='sup_only'):
        super().__init__()
        self.backbone = OracleViT(**backbone)
        self.decode_head = OracleHead(**decode_head)
        self.auxiliary_head = nn.ModuleList([OracleHead(**c) for c in (auxiliary_head or [])])
        self.ema, self.momentum = ema, ema_momentum
        self.unsup_weight, self.unsup_confidence = unsup_weight, unsup_confidence
        self.attn_mask_seperate_head, self.attn_mask_weight = attn_mask_seperate_head, attn_mask_weight
        self.adaptive_attn_mask, self.fdrop_loss_weight, self.patchsize = adaptive_attn_mask, fdrop_loss_weight, patchsize
        self.plain_mt_pseudo_loss = plain_mt_pseudo_loss
        self.use_PatchShuffle_w_Cutmix, self.PatchMix_N, self.patchmix_ratio = use_PatchShuffle_w_Cutmix, PatchMix_N, patchmix_ratio
        self.strong_aug_prob, self.cutout_area = strong_aug_prob, cutout_area
        self.negative_class_ranking = negative_class_ranking and negative_class_ranking_mode in ('unsup_only', 'both')
        assert not negative_class_ranking or negative_class_ranking_mode == 'unsup_only', 'oracle restates mode unsup_only'
        if ema:
            self.backbone_ema = OracleViT(**backbone)
            self.decode_head_ema = OracleHead(**decode_head)
            for p in list(self.backbone_ema.parameters()) + list(self.decode_head_ema.parameters()):
                p.detach_()

    @staticmethod
    def update_ema_variables(model, ema_model, momentum):
        """encoder_decoder.py:1044-1066"""
        for (_, s), (_, t) in zip(model.named_parameters(), ema_model.named_parameters()):
            O.ema_update(t.data, s.data, momentum)
        for (sn, s), (_, t) in zip(model.named_buffers(), ema_model.named_buffers()):
            if 'bn' in sn and 'num_batches_tracked' not in sn:
                O.ema_update(t.data, s.data, momentum)

    def _patch_u(self, conf_mask):
        ps = self.patchsize
        c = conf_mask.view(conf_mask.size(0), conf_mask.size(1) // ps, ps, conf_mask.size(1) // ps, ps)
        c = (1 - c).permute(0, 1, 3, 2, 4)
        c = c.reshape(c.size(0), c.size(1), c.size(2), -1)
        return torch.sum(c, -1) / (ps * ps)

    def teacher_info(self, img):
        """encoder_decoder.py:875-904 (+ :541-542)"""
        with torch.no_grad():
            self.backbone_ema.eval(); self.decode_head_ema.eval()
            feat = self.backbone_ema(img)
            seg_logits = self.decode_head_ema(feat)
            label, conf = O.pseudo_label(seg_logits, self.unsup_confidence)
            self.backbone_ema.train(); self.decode_head_ema.train()
        return dict(seg_logits=seg_logits, hard_seg_label=label, conf_mask=conf)

    def simple_test(self, img, img_shape, ori_shape, flip=None, ema=False):
        """encoder_decoder.py:265-295, 1118-1223 with the intended semantics of mode='whole' (Q8): eval-mode network, logits at
        the input size, then ops.whole_inference_post.  -> (prob, label)"""
        bb, hd = (self.backbone_ema, self.decode_head_ema) if ema else (self.backbone, self.decode_head)
        was = self.training
        self.eval()
        with torch.no_grad():
            out = hd(bb(img))
            if tuple(out.shape[2:]) != tuple(img.shape[2:]):
                out = O.resize(out, img.shape[2:], False)
            res = O.whole_inference_post(out, img_shape, ori_shape, flip, False)
        self.train(was)
        return res

    def slide_test(self, img, img_shape, ori_shape, crop_size, stride, flip=None, ema=False):
        """encoder_decoder.py:1068-1116 (slide_inference) + 1193-1216: overlapping crop_size windows at `stride`, the last one
        of a row / column shifted back inside the image; window logits summed (F.pad to the image = add inside the window),
        divided by the cover count, then ops.whole_inference_post.  -> (prob, label)"""
        bb, hd = (self.backbone_ema, self.decode_head_ema) if ema else (self.backbone, self.decode_head)
        was = self.training
        self.eval()
        with torch.no_grad():
            h_stride, w_stride = stride
            h_crop, w_crop = crop_size
            bs, _, h_img, w_img = img.shape
            h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
            w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
            preds = img.new_zeros((bs, hd.num_classes, h_img, w_img))
            count = img.new_zeros((bs, 1, h_img, w_img))
            for h_idx in range(h_grids):
                for w_idx in range(w_grids):
                    y1, x1 = h_idx * h_stride, w_idx * w_stride
                    y2, x2 = min(y1 + h_crop, h_img), min(x1 + w_crop, w_img)
                    y1, x1 = max(y2 - h_crop, 0), max(x2 - w_crop, 0)
                    crop = img[:, :, y1:y2, x1:x2]
                    out = hd(bb(crop))
                    if tuple(out.shape[2:]) != tuple(crop.shape[2:]):
                        out = O.resize(out, crop.shape[2:], False)
                    preds += F.pad(out, (int(x1), int(w_img - x2), int(y1), int(h_img - y2)))
                    count[:, :, y1:y2, x1:x2] += 1
            assert (count == 0).sum() == 0
            res = O.whole_inference_post(preds / count, img_shape, ori_shape, flip, False)
        self.train(was)
        return res

    def compute_pseudo_loss(self, feat, tinfo, patchmix_n=0, perms=None):
        """encoder_decoder.py:906-954 (NCR: mode 'unsup_only')"""
        pred = self.decode_head(feat, patchmix_n, perms)
        loss = O.ce_none(pred, tinfo['hard_seg_label'], 255)
        mask_ratio = torch.sum(tinfo['conf_mask']).float() / torch.sum(torch.ones_like(loss))
        out = dict(loss_seg_unsup=torch.mean(loss * torch.ones_like(loss)), mask_ratio=mask_ratio)
        if self.negative_class_ranking:
            out['loss_ncr_unsup'] = O.ncr_unsup_only(pred, tinfo['seg_logits'], tinfo['hard_seg_label'])
        return out

    def forward_train(self, img, tags, gt_semantic_seg):
        losses = OrderedDict()
        if self.ema:
            with torch.no_grad():
                self.update_ema_variables(self.backbone, self.backbone_ema, self.momentum)
                self.update_ema_variables(self.decode_head, self.decode_head_ema, self.momentum)
        idx = {t: [i for i, x in enumerate(tags) if x == t] for t in dict.fromkeys(tags)}
        if 'sup' in idx:
            sup_img, sup_gt = img[idx['sup']], gt_semantic_seg[idx['sup']]
            feat = self.backbone(sup_img)
            dec = self.decode_head.forward_train(feat, sup_gt)
            for i, a in enumerate(self.auxiliary_head):
                losses[f'aux_{i}.loss_ce'] = a.forward_train(feat, sup_gt)['loss_ce']
            losses['decode.loss_ce'] = dec['loss_ce']
        self.last = {}
        if 'unsup_student' in idx and self.unsup_weight != 0:
            unsup = OrderedDict()
            tinfo = self.teacher_info(img[idx['unsup_teacher']])
            self.last['teacher'] = tinfo
            simg = img[idx['unsup_student']]
            if self.attn_mask_seperate_head:
                u = self._patch_u(tinfo['conf_mask'])
                feat = self.backbone(simg, attn_mask=u, attn_mask_weight=self.attn_mask_weight,
                                     adaptive_attn_mask=self.adaptive_attn_mask)
                unsup['loss_seg_unsup_attn_mask'] = self.compute_pseudo_loss(feat, tinfo)['loss_seg_unsup'] * 0.5
                pm_n, perms = 0, None
                if self.use_PatchShuffle_w_Cutmix:
                    # encoder_decoder.py:633-638: CutMix (images + pseudo-labels; conf_mask and seg_logits stay), then
                    # PatchShuffle of the images; the decode head un-shuffles the tokens (labels stay in place)
                    tinfo = dict(tinfo)
                    boxes, perms = O.draw_strong_aug(simg.shape[0], tuple(simg.shape[2:]), self.strong_aug_prob, self.cutout_area,
                                                     self.patchmix_ratio, self.patchsize * self.PatchMix_N)
                    simg, tinfo['hard_seg_label'] = O.cutmix(simg, tinfo['hard_seg_label'], boxes)
                    simg = O.patch_shuffle(simg, perms, self.patchsize * self.PatchMix_N)
                    pm_n = self.PatchMix_N
                feat = self.backbone(simg)
                r = self.compute_pseudo_loss(feat, tinfo, pm_n, perms)
                self.last['mask_ratio'] = r['mask_ratio']
                if self.negative_class_ranking:
                    unsup['loss_ncr_unsup'] = r['loss_ncr_unsup'] * 0.5
                unsup['loss_seg_unsup'] = r['loss_seg_unsup'] * self.fdrop_loss_weight
            else:
                u = self._patch_u(tinfo['conf_mask'])
                feat = self.backbone(simg, attn_mask=u, attn_mask_weight=self.attn_mask_weight,
                                     adaptive_attn_mask=self.adaptive_attn_mask)
            if self.plain_mt_pseudo_loss and not self.attn_mask_seperate_head:
                r = self.compute_pseudo_loss(feat, tinfo)
                self.last['mask_ratio'] = r['mask_ratio']
                unsup['loss_seg_unsup'] = r['loss_seg_unsup'] * self.fdrop_loss_weight
            for k in unsup:
                if 'loss' in k:
                    unsup[k] = unsup[k] * self.unsup_weight
            losses.update(unsup)
        return losses

    @staticmethod
    def parse_losses(losses):
        """base.py:230-274 (single process)"""
        log_vars = OrderedDict((k, v.mean()) for k, v in losses.items())
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        return loss, log_vars


def build_optimizer(model, lr, momentum=0.9, head_mult=10.0):
    """torch.optim.SGD with one group per parameter, lr x10 for names containing 'head'
    (mmcv DefaultOptimizerConstructor, configs/setr/*_sup.py:244-247)."""
    groups = []
    for n, p in model.named_parameters():
        g = {'params': [p]}
        if p.requires_grad and 'head' in n:
            g['lr'] = lr * head_mult
        groups.append(g)
    opt = torch.optim.SGD(groups, lr=lr, momentum=momentum, weight_decay=0.0)
    for g in opt.param_groups:
        g['initial_lr'] = g['lr']
    return opt


def set_poly_lr(opt, it, max_iters=80001, power=0.9, min_lr=1e-4):
    for g in opt.param_groups:
        g['lr'] = O.poly_lr(g['initial_lr'], it, max_iters, power, min_lr)


def oracle_from_cfg(model_cfg):
    """build the oracle from an mmseg-style model dict (the keys the hot path uses)"""
    def bb(c):
        return dict(img_size=tuple(c['img_size']), patch_size=c.get('patch_size', 16), in_channels=c.get('in_channels', 3),
                    embed_dims=c.get('embed_dims', 768), num_layers=c.get('num_layers', 12), num_heads=c.get('num_heads', 12),
                    mlp_ratio=c.get('mlp_ratio', 4), out_indices=tuple(c.get('out_indices', (11,))),
                    eps=c.get('norm_cfg', {}).get('eps', 1e-5))

    def hd(c):
        return dict(in_channels=c['in_channels'], channels=c['channels'], num_classes=c['num_classes'],
                    num_convs=c.get('num_convs', 1), up_scale=c.get('up_scale', 4), in_index=c.get('in_index', -1),
                    loss_weight=c.get('loss_decode', {}).get('loss_weight', 1.0),
                    ln_eps=c.get('norm_layer', dict(eps=1e-6)).get('eps', 1e-6))

    aux = model_cfg.get('auxiliary_head')
    return OracleSegmentor(
        bb(model_cfg['backbone']), hd(model_cfg['decode_head']), [hd(a) for a in aux] if aux else None,
        ema=model_cfg.get('ema', False), ema_momentum=model_cfg.get('ema_momentum', 0.999),
        unsup_weight=model_cfg.get('unsup_weight', 2.0), unsup_confidence=model_cfg.get('unsup_confidence', 0.75),
        attn_mask_seperate_head=model_cfg.get('attn_mask_seperate_head', False),
        attn_mask_weight=model_cfg.get('attn_mask_weight', 50), adaptive_attn_mask=model_cfg.get('adaptive_attn_mask', False),
        fdrop_loss_weight=model_cfg.get('fdrop_loss_weight', 0.5), patchsize=model_cfg.get('patchsize', 16),
        plain_mt_pseudo_loss=model_cfg.get('plain_mt_pseudo_loss', False),
        use_PatchShuffle_w_Cutmix=model_cfg.get('use_PatchShuffle_w_Cutmix', False), PatchMix_N=model_cfg.get('PatchMix_N', 8),
        patchmix_ratio=model_cfg.get('patchmix_ratio', 0.5), strong_aug_prob=model_cfg.get('strong_aug_prob', 0.5),
        cutout_area=model_cfg.get('cutout_area', 2), negative_class_ranking=model_cfg.get('negative_class_ranking', False),
        negative_class_ranking_mode=model_cfg.get('negative_class_ranking_mode', 'sup_only'))
