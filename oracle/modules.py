"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement, in plain torch fp32 ops, of the reference operator library on the detect hot path
(SURVEY.md §8a rows 1-2, 4-11, 14-18).  Every class cites the reference file:line it restates
(paths relative to /root/reference/ultralytics/).  Parameter / buffer names reproduce the reference
state_dict contract (§8a row 0) so procedural weights keyed by name land on the same tensors.

Pinned by tests/golden/*.npz, which gen_golden.py produced by running the imported reference itself.
"""

from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def autopad(k, p=None, d=1):
    """'same' padding (nn/modules/conv.py:64-70)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class Conv(nn.Module):
    """conv2d(bias=False) -> BatchNorm2d -> SiLU (nn/modules/conv.py:147-197)."""

    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))

    def forward_fuse(self, x):
        return self.act(self.conv(x))


def fuse_conv_and_bn(conv: nn.Conv2d, bn: nn.BatchNorm2d) -> nn.Conv2d:
    """W' = diag(g/sqrt(var+eps)) W ; b' = beta - g*mu/sqrt(var+eps) (+ scaled conv bias)
    (utils/torch_utils.py:236-266; same operation order so fp32 bits agree with the reference)."""
    w_conv = conv.weight.view(conv.out_channels, -1)
    w_bn = torch.diag(bn.weight.div(torch.sqrt(bn.eps + bn.running_var)))
    conv.weight.data = torch.mm(w_bn, w_conv).view(conv.weight.shape)
    b_conv = torch.zeros(conv.out_channels) if conv.bias is None else conv.bias
    b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
    fused_bias = torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn
    if conv.bias is None:
        conv.register_parameter("bias", nn.Parameter(fused_bias))
    else:
        conv.bias.data = fused_bias
    return conv.requires_grad_(False)


class Concat(nn.Module):
    """torch.cat along `dimension` (nn/modules/conv.py:850-875)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


class Bottleneck(nn.Module):
    """x + cv2(cv1(x)) when shortcut and c1 == c2 (nn/modules/block.py:644-668)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C2f(nn.Module):
    """cv1 -> chunk(2) -> n chained Bottlenecks -> cat(2+n) -> cv2 (nn/modules/block.py:457-488)."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        y.extend(m(y[-1]) for m in self.m)
        return self.cv2(torch.cat(y, 1))


class C3(nn.Module):
    """cv3(cat(m(cv1(x)), cv2(x))), m = n x Bottleneck(k=(1,3), e=1) (nn/modules/block.py:509-532)."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=((1, 1), (3, 3)), e=1.0) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class SPPF(nn.Module):
    """cv1 -> 3 chained MaxPool2d(5,1,2) -> cat 4 -> cv2 (nn/modules/block.py:382-406)."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        y = [self.cv1(x)]
        y.extend(self.m(y[-1]) for _ in range(3))
        return self.cv2(torch.cat(y, 1))


class DFL(nn.Module):
    """softmax over 16 bins . arange(16) (nn/modules/block.py:232-253)."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1

    def forward(self, x):
        b, _, a = x.shape
        return self.conv(x.view(b, 4, self.c1, a).transpose(2, 1).softmax(1)).view(b, 4, a)


def make_anchors(feats, strides, grid_cell_offset=0.5):
    """Cell-centre anchor points + per-anchor stride (utils/tal.py:352-364)."""
    pts, st = [], []
    for i, stride in enumerate(strides):
        h, w = feats[i].shape[2:]
        sx = torch.arange(end=w, dtype=torch.float32) + grid_cell_offset
        sy = torch.arange(end=h, dtype=torch.float32) + grid_cell_offset
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(stride), dtype=torch.float32))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):
    """ltrb distances -> xywh / xyxy (utils/tal.py:367-376)."""
    lt, rb = distance.chunk(2, dim)
    x1y1 = anchor_points - lt
    x2y2 = anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


class Detect(nn.Module):
    """YOLO Detect head, legacy (v3/v5/v8) cls branch (nn/modules/head.py:28-191)."""

    legacy = True  # parse_model leaves legacy=True for the v3/v5/v8 YAMLs (nn/tasks.py:2424,2994)

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(
            nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(
            nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max)

    def forward(self, x):
        x = list(x)
        for i in range(self.nl):
            x[i] = torch.cat((self.cv2[i](x[i]), self.cv3[i](x[i])), 1)
        if self.training:
            return x
        return self._inference(x), x

    def _inference(self, x):
        """head.py:151-169: cat levels, DFL, dist2bbox * stride, sigmoid(cls)."""
        b = x[0].shape[0]
        x_cat = torch.cat([xi.view(b, self.no, -1) for xi in x], 2)
        anchors, strides = (t.transpose(0, 1) for t in make_anchors(x, self.stride, 0.5))
        box, cls = x_cat.split((self.reg_max * 4, self.nc), 1)
        dbox = dist2bbox(self.dfl(box), anchors.unsqueeze(0), xywh=True, dim=1) * strides
        return torch.cat((dbox, cls.sigmoid()), 1)

    def bias_init(self):
        """head.py:171-178."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)


class MHSA(nn.Module):
    """4-head dense self attention on a feature map, unscaled q^T k, no positional term
    (nn/modules/block.py:6020-6062)."""

    def __init__(self, n_dims, width=14, height=14, heads=4, pos_emb=False):
        super().__init__()
        assert not pos_emb, "pos_emb is never enabled on the hot path (block.py:6078)"
        self.heads = heads
        self.query = nn.Conv2d(n_dims, n_dims, kernel_size=1)
        self.key = nn.Conv2d(n_dims, n_dims, kernel_size=1)
        self.value = nn.Conv2d(n_dims, n_dims, kernel_size=1)

    def forward(self, x):
        b, c, w, h = x.size()
        q = self.query(x).view(b, self.heads, c // self.heads, -1)
        k = self.key(x).view(b, self.heads, c // self.heads, -1)
        v = self.value(x).view(b, self.heads, c // self.heads, -1)
        energy = torch.matmul(q.permute(0, 1, 3, 2), k)
        attention = energy.softmax(-1)
        out = torch.matmul(v, attention.permute(0, 1, 3, 2))
        return out.view(b, c, w, h)


class BottleneckTransformer(nn.Module):
    """x + MHSA(cv1(x)); `fc1` is a dead parameter kept for the state_dict (block.py:6065-6092)."""

    def __init__(self, c1, c2, stride=1, heads=4, mhsa=True, resolution=None, expansion=1):
        super().__init__()
        assert mhsa and stride == 1 and c1 == expansion * c2
        c_ = int(c2 * expansion)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = nn.Sequential(MHSA(c2, width=int(resolution[0]), height=int(resolution[1]), heads=heads))
        self.shortcut = c1 == c2
        self.fc1 = nn.Linear(c2, c2)

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.shortcut else self.cv2(self.cv1(x))


class BoT3(nn.Module):
    """CSP block whose inner blocks are BottleneckTransformers (block.py:6095-6109)."""

    def __init__(self, c1, c2, n=1, e=0.5, e2=1, w=20, h=20):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(
            *[BottleneckTransformer(c_, c_, stride=1, heads=4, mhsa=True, resolution=(w, h), expansion=e2)
              for _ in range(n)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))


# ---------------------------------------------------------------------------------------------------------------------
# RT-DETR decoder head (config 5)
# ---------------------------------------------------------------------------------------------------------------------
def inverse_sigmoid(x, eps=1e-5):
    """log(clamp(x)/clamp(1-x)) (nn/modules/utils.py:79-100)."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def bias_init_with_prob(prior_prob=0.01):
    """nn/modules/utils.py:35-51."""
    return float(-math.log((1 - prior_prob) / prior_prob))


class MLP(nn.Module):
    """Linear/ReLU stack (nn/modules/transformer.py:348-399)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim, *h], [*h, output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x


def multi_scale_deformable_attn(value, shapes, sampling_locations, attention_weights):
    """Bilinear multi-scale sampling + weighted sum (nn/modules/utils.py:103-159)."""
    bs, _, nh, hd = value.shape
    _, nq, _, nl, npts, _ = sampling_locations.shape
    value_list = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        v = value_list[lvl].flatten(2).transpose(1, 2).reshape(bs * nh, hd, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(bs * nh, 1, nq, nl * npts)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(bs, nh * hd, nq)
    return out.transpose(1, 2).contiguous()


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention (nn/modules/transformer.py:438-558); 4-d reference boxes branch :552-554."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """transformer.py:487-508."""
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(
            1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, refer_bbox, value, value_shapes, value_mask=None):
        bs, len_q = query.shape[:2]
        len_v = value.shape[1]
        value = self.value_proj(value)
        if value_mask is not None:
            value = value.masked_fill(value_mask[..., None], float(0))
        value = value.view(bs, len_v, self.n_heads, self.d_model // self.n_heads)
        so = self.sampling_offsets(query).view(bs, len_q, self.n_heads, self.n_levels, self.n_points, 2)
        aw = self.attention_weights(query).view(bs, len_q, self.n_heads, self.n_levels * self.n_points)
        aw = F.softmax(aw, -1).view(bs, len_q, self.n_heads, self.n_levels, self.n_points)
        npts = refer_bbox.shape[-1]
        if npts == 2:
            norm = torch.as_tensor(value_shapes, dtype=query.dtype).flip(-1)
            loc = refer_bbox[:, :, None, :, None, :] + so / norm[None, None, None, :, None, :]
        else:
            add = so / self.n_points * refer_bbox[:, :, None, :, None, 2:] * 0.5
            loc = refer_bbox[:, :, None, :, None, :2] + add
        return self.output_proj(multi_scale_deformable_attn(value, value_shapes, loc, aw))


class DeformableTransformerDecoderLayer(nn.Module):
    """self-attn(MHA, seq-first) + MSDeformAttn + FFN with post-LayerNorms (transformer.py:561-685); dropout 0."""

    def __init__(self, d_model=256, n_heads=8, d_ffn=1024, dropout=0.0, act=None, n_levels=4, n_points=4):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm3 = nn.LayerNorm(d_model)

    def forward(self, embed, refer_bbox, feats, shapes, padding_mask=None, attn_mask=None, query_pos=None):
        q = k = embed if query_pos is None else embed + query_pos
        tgt = self.self_attn(q.transpose(0, 1), k.transpose(0, 1), embed.transpose(0, 1),
                             attn_mask=attn_mask)[0].transpose(0, 1)
        embed = self.norm1(embed + tgt)
        tgt = self.cross_attn(embed if query_pos is None else embed + query_pos, refer_bbox.unsqueeze(2), feats,
                              shapes, padding_mask)
        embed = self.norm2(embed + tgt)
        tgt2 = self.linear2(F.relu(self.linear1(embed)))
        return self.norm3(embed + tgt2)


class DeformableTransformerDecoder(nn.Module):
    """6 layers, iterative box refinement, eval exit at eval_idx (transformer.py:688-773)."""

    def __init__(self, hidden_dim, decoder_layer, num_layers, eval_idx=-1):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.eval_idx = eval_idx if eval_idx >= 0 else num_layers + eval_idx

    def forward(self, embed, refer_bbox, feats, shapes, bbox_head, score_head, pos_mlp, attn_mask=None,
                padding_mask=None):
        output = embed
        dec_bboxes, dec_cls = [], []
        refer_bbox = refer_bbox.sigmoid()
        for i, layer in enumerate(self.layers):
            output = layer(output, refer_bbox, feats, shapes, padding_mask, attn_mask, pos_mlp(refer_bbox))
            bbox = bbox_head[i](output)
            refined = torch.sigmoid(bbox + inverse_sigmoid(refer_bbox))
            if i == self.eval_idx:
                dec_cls.append(score_head[i](output))
                dec_bboxes.append(refined)
                break
            refer_bbox = refined
        return torch.stack(dec_bboxes), torch.stack(dec_cls)


class RTDETRDecoder(nn.Module):
    """RT-DETR query selection + deformable decoder, eval path only (nn/modules/head.py:1905-2224).
    `get_cdn_group` returns 4 x None in eval (models/utils/ops.py:232-233), so the denoising branch is omitted;
    `denoising_class_embed` is kept for the state_dict."""

    def __init__(self, nc=80, ch=(512, 1024, 2048), hd=256, nq=300, ndp=4, nh=8, ndl=6, d_ffn=1024, dropout=0.0,
                 act=None, eval_idx=-1, nd=100, label_noise_ratio=0.5, box_noise_scale=1.0, learnt_init_query=False):
        super().__init__()
        assert not learnt_init_query
        self.hidden_dim, self.nhead, self.nl, self.nc = hd, nh, len(ch), nc
        self.num_queries, self.num_decoder_layers = nq, ndl
        self.input_proj = nn.ModuleList(nn.Sequential(nn.Conv2d(x, hd, 1, bias=False), nn.BatchNorm2d(hd)) for x in ch)
        layer = DeformableTransformerDecoderLayer(hd, nh, d_ffn, dropout, act, self.nl, ndp)
        self.decoder = DeformableTransformerDecoder(hd, layer, ndl, eval_idx)
        self.denoising_class_embed = nn.Embedding(nc, hd)
        self.query_pos_head = MLP(4, 2 * hd, hd, num_layers=2)
        self.enc_output = nn.Sequential(nn.Linear(hd, hd), nn.LayerNorm(hd))
        self.enc_score_head = nn.Linear(hd, nc)
        self.enc_bbox_head = MLP(hd, hd, 4, num_layers=3)
        self.dec_score_head = nn.ModuleList([nn.Linear(hd, nc) for _ in range(ndl)])
        self.dec_bbox_head = nn.ModuleList([MLP(hd, hd, 4, num_layers=3) for _ in range(ndl)])

    @staticmethod
    def _generate_anchors(shapes, grid_size=0.05, eps=1e-2):
        """head.py:2078-2115."""
        anchors = []
        for i, (h, w) in enumerate(shapes):
            sy = torch.arange(end=h, dtype=torch.float32)
            sx = torch.arange(end=w, dtype=torch.float32)
            gy, gx = torch.meshgrid(sy, sx, indexing="ij")
            gxy = (torch.stack([gx, gy], -1).unsqueeze(0) + 0.5) / torch.tensor([w, h], dtype=torch.float32)
            wh = torch.ones_like(gxy) * grid_size * (2.0 ** i)
            anchors.append(torch.cat([gxy, wh], -1).view(-1, h * w, 4))
        anchors = torch.cat(anchors, 1)
        valid = ((anchors > eps) & (anchors < 1 - eps)).all(-1, keepdim=True)
        anchors = torch.log(anchors / (1 - anchors)).masked_fill(~valid, float("inf"))
        return anchors, valid

    def forward(self, x, batch=None):
        # _get_encoder_input (head.py:2117-2141)
        proj = [self.input_proj[i](f) for i, f in enumerate(x)]
        shapes = [[int(p.shape[2]), int(p.shape[3])] for p in proj]
        feats = torch.cat([p.flatten(2).permute(0, 2, 1) for p in proj], 1)
        # _get_decoder_input (head.py:2143-2200)
        bs = feats.shape[0]
        anchors, valid = self._generate_anchors(shapes)
        features = self.enc_output(valid * feats)
        scores = self.enc_score_head(features)
        topk = torch.topk(scores.max(-1).values, self.num_queries, dim=1).indices.view(-1)
        bidx = torch.arange(bs).unsqueeze(-1).repeat(1, self.num_queries).view(-1)
        top_feat = features[bidx, topk].view(bs, self.num_queries, -1)
        top_anch = anchors[:, topk].view(bs, self.num_queries, -1)
        refer_bbox = self.enc_bbox_head(top_feat) + top_anch
        enc_bboxes = refer_bbox.sigmoid()
        enc_scores = scores[bidx, topk].view(bs, self.num_queries, -1)
        dec_bboxes, dec_scores = self.decoder(top_feat, refer_bbox, feats, shapes, self.dec_bbox_head,
                                              self.dec_score_head, self.query_pos_head)
        y = torch.cat((dec_bboxes.squeeze(0), dec_scores.squeeze(0).sigmoid()), -1)
        return y, (dec_bboxes, dec_scores, enc_bboxes, enc_scores, None)
