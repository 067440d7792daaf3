"""RT-DETR decoder head with the reference's operator API (ultralytics/nn/modules/head.py:1905-2224 and
nn/modules/transformer.py:348-773), eval path, on HIP kernels.

Module / parameter names reproduce the reference state_dict (input_proj, decoder.layers.N.{self_attn,cross_attn,...},
enc_output, enc_score_head, enc_bbox_head, dec_score_head, dec_bbox_head, query_pos_head, denoising_class_embed).
The classes below are parameter containers + an orchestration of upa_* launches; tokens are f32 rows stored
level-major (all images of level 0, then level 1, ...), which makes every level a plain NHWC view for the input
projection and for the deformable sampling, and lets every nn.Linear run as one MFMA GEMM over all tokens.
"""

from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn

from ... import _lib as L
from ...engine import runtime as R
from .conv import PackedConv, fold_bn, hip_conv2d, version_key

__all__ = ("RTDETRDecoder", "MLP", "MSDeformAttn", "DeformableTransformerDecoderLayer", "DeformableTransformerDecoder")


# ---------------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------------
class _Rows:
    """A (rows, cols) float32 device matrix (contiguous)."""

    @staticmethod
    def new(m, c, device, key=None):
        return R.alloc_plain((m, c), torch.float32, device, key=key)


# Perf mode of the decoder's nn.Linear layers (set by RTDETRDecoder.forward while a bf16 backbone feeds it): the product runs on the bf16
# matrix cores (upa_linear_bf16: x rounded to bf16 on the way into the MFMA, bf16 weights, float32 accumulate / bias / activation /
# residual / storage) - the arithmetic of the reference's half-precision predict, where the whole RTDETRDecoder is `.half()`.  In
# parity mode (float32 backbone) every product stays exact float32.
_LINEAR_BF16 = [False]


def _packed_linear(owner: nn.Module, name: str, weight: torch.Tensor, bias, device, dtype=torch.float32) -> PackedConv:
    """Packed GEMM weights (float32, or bf16 for the perf mode) of one nn.Linear, cached on `owner` and validated against the
    parameters' storage / in-place version on every use (`version_key`): load_state_dict / load_weights / .to() can never leave a
    stale copy."""
    cache = owner.__dict__.setdefault("_pk_cache", {})
    key = (name, str(device)) if dtype == torch.float32 else (name, str(device), "bf16")
    ver = version_key(weight, bias)
    hit = cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    w = weight.detach().float().cpu().reshape(weight.shape[0], weight.shape[1], 1, 1)
    b = torch.zeros(weight.shape[0]) if bias is None else bias.detach().float().cpu()
    pk = PackedConv(w, b, 1, device, dtype, False)
    cache[key] = (ver, pk)
    return pk


def linear(owner, name, weight, bias, x, act=L.ACT_NONE, residual=None, key=None):
    """y = act(x W^T + b) (+ residual): exact float32 MFMA GEMM (upa_linear), or in the decoder's perf mode the bf16-product form
    (upa_linear_bf16) wherever the shape allows it."""
    m, k = x.shape
    st = L.current_stream(x.device)
    rp, rld = (None, 0) if residual is None else (residual.data_ptr(), residual.stride(0))
    if _LINEAR_BF16[0] and k % 32 == 0 and k <= 1024 and weight.shape[0] % 4 == 0:
        pk = _packed_linear(owner, name, weight, bias, x.device, torch.bfloat16)
        y = _Rows.new(m, pk.cout, x.device, key=key)
        rc = L.lib().upa_linear_bf16(x.data_ptr(), m, k, x.stride(0), pk.w.data_ptr(), pk.bias.data_ptr(), y.data_ptr(), pk.cout,
                                     y.stride(0), rp, rld, act, st)
        if rc != L.UPA_EUNSUPPORTED:
            L.check(rc, "linear_bf16")
            return y
    pk = _packed_linear(owner, name, weight, bias, x.device)
    y = _Rows.new(m, pk.cout, x.device, key=key)
    L.check(L.lib().upa_linear(x.data_ptr(), m, k, x.stride(0), pk.w.data_ptr(), pk.bias.data_ptr(), y.data_ptr(), pk.cout,
                               y.stride(0), rp, rld, act, st), "linear")
    return y


def layer_norm(ln: nn.LayerNorm, x, residual=None, key=None):
    m, c = x.shape
    y = _Rows.new(m, c, x.device, key=key)
    L.check(L.lib().upa_layer_norm(x.data_ptr(), None if residual is None else residual.data_ptr(), m, c,
                                   ln.weight.data_ptr(), ln.bias.data_ptr(), float(ln.eps), y.data_ptr(),
                                   L.current_stream(x.device)), "layer_norm")
    return y


def rows_add(a, b, key=None):
    y = _Rows.new(a.shape[0], a.shape[1], a.device, key=key)
    L.check(L.lib().upa_rows_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.shape[0], a.shape[1],
                                 L.current_stream(a.device)), "rows_add")
    return y


# ---------------------------------------------------------------------------------------------------------------------
# modules (parameter containers with HIP forward helpers)
# ---------------------------------------------------------------------------------------------------------------------
class MLP(nn.Module):
    """Linear/ReLU stack (transformer.py:348-399)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, act=nn.ReLU, sigmoid=False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim, *h], [*h, output_dim]))
        self.sigmoid = sigmoid
        self.act = act()

    def forward(self, x, key=None):
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            x = linear(self, f"l{i}", layer.weight, layer.bias, x, L.ACT_NONE if last else L.ACT_RELU,
                       key=(key, id(self), i))
        return x


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention (transformer.py:438-558)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        self.im2col_step = 64
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

    def forward(self, query, refer_bbox, value, shapes_dev, bs, residual=None, key=None, projected=None):
        """query (bs*nq, C) rows, refer_bbox (bs*nq, 4), value = level-major token rows (sum_l bs*H_l*W_l, C).
        projected = (pointer, dtype code, row stride): this layer's value projection already computed (all layers' value
        projections as one bf16 GEMM, RTDETRDecoder.forward in perf mode)."""
        nq = query.shape[0] // bs
        if projected is None:
            val = linear(self, "value_proj", self.value_proj.weight, self.value_proj.bias, value, key=(key, "val"))
            projected = (val.data_ptr(), L.UPA_F32, self.d_model)
        off = linear(self, "sampling_offsets", self.sampling_offsets.weight, self.sampling_offsets.bias, query,
                     key=(key, "off"))
        aw = linear(self, "attention_weights", self.attention_weights.weight, self.attention_weights.bias, query,
                    key=(key, "aw"))
        samp = _Rows.new(query.shape[0], self.d_model, query.device, key=(key, "samp"))
        L.check(L.lib().upa_msdeform_attn_strided(projected[0], projected[1], projected[2], shapes_dev["host_ptr"], self.n_levels,
                                                  bs, self.n_heads, self.d_model // self.n_heads, off.data_ptr(), aw.data_ptr(),
                                                  refer_bbox.data_ptr(), nq, self.n_points, samp.data_ptr(),
                                                  L.current_stream(query.device)), "msdeform_attn")
        return linear(self, "output_proj", self.output_proj.weight, self.output_proj.bias, samp, residual=residual,
                      key=(key, "out"))


class DeformableTransformerDecoderLayer(nn.Module):
    """self-attention (nn.MultiheadAttention) + MSDeformAttn + FFN with post-LayerNorms (transformer.py:561-685)."""

    def __init__(self, d_model=256, n_heads=8, d_ffn=1024, dropout=0.0, act=nn.ReLU(), n_levels=4, n_points=4):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.act = act
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    def forward(self, embed, refer_bbox, feats, shapes_dev, bs, query_pos, key=None, projected=None):
        """embed, query_pos: (bs*nq, C) rows.  Returns the updated embed."""
        mha = self.self_attn
        e, nh = mha.embed_dim, mha.num_heads
        nq = embed.shape[0] // bs
        w, b = mha.in_proj_weight, mha.in_proj_bias
        qk_in = rows_add(embed, query_pos, key=(key, "qk_in"))
        # in-projection: q,k from (embed + pos), v from embed (transformer.py:670-673), into one (rows, 3C) buffer
        lib, st = L.lib(), L.current_stream(embed.device)
        d = e // nh
        fast = _LINEAR_BF16[0] and e % 64 == 0 and e <= 1024 and d == 32 and nq >= 16
        if fast:
            # perf mode: q, k, v written as bf16 rows by the bf16-product GEMM, attention on the matrix cores (csrc/attention.hip:
            # mhsa_mfma_bf16_d32_kernel, the BoT3 kernel with nn.MultiheadAttention's 1/sqrt(d)), its bf16 output straight into out_proj
            qkv = R.alloc_plain((embed.shape[0], 3 * e), torch.bfloat16, embed.device, key=(key, "qkv16"))
            pk_qk = _packed_linear(self, "in_qk", w[: 2 * e], b[: 2 * e], embed.device, torch.bfloat16)
            pk_v = _packed_linear(self, "in_v", w[2 * e:], b[2 * e:], embed.device, torch.bfloat16)
            L.check(lib.upa_linear_mixed(qk_in.data_ptr(), L.UPA_F32, qk_in.shape[0], e, e, pk_qk.w.data_ptr(), pk_qk.bias.data_ptr(),
                                         qkv.data_ptr(), L.UPA_BF16, 2 * e, 3 * e, None, 0, L.ACT_NONE, st), "in_proj_qk")
            L.check(lib.upa_linear_mixed(embed.data_ptr(), L.UPA_F32, embed.shape[0], e, e, pk_v.w.data_ptr(), pk_v.bias.data_ptr(),
                                         qkv.data_ptr() + 2 * e * 2, L.UPA_BF16, e, 3 * e, None, 0, L.ACT_NONE, st), "in_proj_v")
            attn = R.alloc_plain((embed.shape[0], e), torch.bfloat16, embed.device, key=(key, "attn16"))
            L.check(lib.upa_mhsa(qkv.data_ptr(), qkv.data_ptr() + e * 2, qkv.data_ptr() + 2 * e * 2, 3 * e, bs, nq, nh, d,
                                 1.0 / math.sqrt(d), None, 0, attn.data_ptr(), e, L.UPA_BF16, st), "self_attn")
            pk_o = _packed_linear(self, "out_proj", mha.out_proj.weight, mha.out_proj.bias, embed.device, torch.bfloat16)
            tgt = _Rows.new(embed.shape[0], e, embed.device, key=(key, "sa_out"))
            L.check(lib.upa_linear_mixed(attn.data_ptr(), L.UPA_BF16, attn.shape[0], e, e, pk_o.w.data_ptr(), pk_o.bias.data_ptr(),
                                         tgt.data_ptr(), L.UPA_F32, e, e, embed.data_ptr(), embed.stride(0), L.ACT_NONE, st), "out_proj")
        else:
            qkv = _Rows.new(embed.shape[0], 3 * e, embed.device, key=(key, "qkv"))
            lin = lib.upa_linear_bf16 if (_LINEAR_BF16[0] and e % 32 == 0 and e <= 1024) else lib.upa_linear
            wdt = torch.bfloat16 if lin is lib.upa_linear_bf16 else torch.float32
            pk_qk = _packed_linear(self, "in_qk", w[: 2 * e], b[: 2 * e], embed.device, wdt)
            pk_v = _packed_linear(self, "in_v", w[2 * e:], b[2 * e:], embed.device, wdt)
            L.check(lin(qk_in.data_ptr(), qk_in.shape[0], e, e, pk_qk.w.data_ptr(), pk_qk.bias.data_ptr(),
                        qkv.data_ptr(), 2 * e, 3 * e, None, 0, L.ACT_NONE, st), "in_proj_qk")
            L.check(lin(embed.data_ptr(), embed.shape[0], e, e, pk_v.w.data_ptr(), pk_v.bias.data_ptr(),
                        qkv.data_ptr() + 2 * e * 4, e, 3 * e, None, 0, L.ACT_NONE, st), "in_proj_v")
            attn = _Rows.new(embed.shape[0], e, embed.device, key=(key, "attn"))
            L.check(lib.upa_mhsa(qkv.data_ptr(), qkv.data_ptr() + e * 4, qkv.data_ptr() + 2 * e * 4, 3 * e, bs, nq, nh, d,
                                 1.0 / math.sqrt(d), None, 0, attn.data_ptr(), e, L.UPA_F32, st), "self_attn")
            tgt = linear(self, "out_proj", mha.out_proj.weight, mha.out_proj.bias, attn, residual=embed, key=(key, "sa_out"))
        embed = layer_norm(self.norm1, tgt, key=(key, "n1"))
        # cross attention
        q2 = rows_add(embed, query_pos, key=(key, "q2"))
        tgt = self.cross_attn(q2, refer_bbox, feats, shapes_dev, bs, residual=embed, key=(key, "ca"), projected=projected)
        embed = layer_norm(self.norm2, tgt, key=(key, "n2"))
        # FFN
        hdn = linear(self, "linear1", self.linear1.weight, self.linear1.bias, embed, L.ACT_RELU, key=(key, "f1"))
        tgt = linear(self, "linear2", self.linear2.weight, self.linear2.bias, hdn, residual=embed, key=(key, "f2"))
        return layer_norm(self.norm3, tgt, key=(key, "n3"))


class DeformableTransformerDecoder(nn.Module):
    """Decoder stack with iterative box refinement and eval exit at eval_idx (transformer.py:688-773)."""

    def __init__(self, hidden_dim, decoder_layer, num_layers, eval_idx=-1):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(decoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        self.eval_idx = eval_idx if eval_idx >= 0 else num_layers + eval_idx

    def forward(self, embed, refer_logit, feats, shapes_dev, bs, bbox_head, score_head, pos_mlp, projected=None):
        """projected: None, or (pointer, dtype code, row stride) of the value projections of ALL layers side by side
        (layer i = columns [i*C, (i+1)*C))."""
        lib, dev = L.lib(), embed.device
        st = L.current_stream(dev)
        m = embed.shape[0]
        refer = _Rows.new(m, 4, dev, key=(id(self), "ref0"))
        L.check(lib.upa_sigmoid(refer_logit.data_ptr(), refer.data_ptr(), m * 4, st), "sigmoid")
        output = embed
        for i, layer in enumerate(self.layers):
            pos = pos_mlp(refer, key=(id(self), "pos", i))
            pj = None
            if projected is not None:
                es = 2 if projected[1] == L.UPA_BF16 else 4
                pj = (projected[0] + i * self.hidden_dim * es, projected[1], projected[2])
            output = layer(output, refer, feats, shapes_dev, bs, pos, key=(id(self), i), projected=pj)
            bbox = bbox_head[i](output, key=(id(self), "bb", i))
            refined = _Rows.new(m, 4, dev, key=(id(self), "ref", i + 1))
            L.check(lib.upa_box_refine(bbox.data_ptr(), refer.data_ptr(), refined.data_ptr(), m, st), "box_refine")
            if i == self.eval_idx:
                sc = score_head[i]
                scores = linear(self, f"score{i}", sc.weight, sc.bias, output, key=(id(self), "score"))
                return refined, scores
            refer = refined
        raise L.UpaError("eval_idx beyond the number of decoder layers")


class RTDETRDecoder(nn.Module):
    """Real-Time Deformable Transformer Decoder head, inference path (head.py:1905-2224).

    forward(x: [P3, P4, P5] NHWC views) -> (y, aux) with y = (bs, num_queries, 4 + nc): normalised cxcywh boxes and
    class probabilities, like the reference's eval return (head.py:2073-2075)."""

    export = False
    shapes = []
    anchors = torch.empty(0)
    valid_mask = torch.empty(0)
    dynamic = False
    fuse_value_proj = True  # perf mode: one bf16 GEMM for the value projections of all decoder layers

    def __init__(self, nc=80, ch=(512, 1024, 2048), hd=256, nq=300, ndp=4, nh=8, ndl=6, d_ffn=1024, dropout=0.0,
                 act=nn.ReLU(), eval_idx=-1, nd=100, label_noise_ratio=0.5, box_noise_scale=1.0, learnt_init_query=False):
        super().__init__()
        if learnt_init_query:
            raise L.UpaError("learnt_init_query is not used by the reference YAMLs on the hot path")
        self.hidden_dim, self.nhead, self.nl, self.nc = hd, nh, len(ch), nc
        self.num_queries, self.num_decoder_layers = nq, ndl
        self.input_proj = nn.ModuleList(nn.Sequential(nn.Conv2d(x, hd, 1, bias=False), nn.BatchNorm2d(hd)) for x in ch)
        decoder_layer = DeformableTransformerDecoderLayer(hd, nh, d_ffn, dropout, act, self.nl, ndp)
        self.decoder = DeformableTransformerDecoder(hd, decoder_layer, ndl, eval_idx)
        self.denoising_class_embed = nn.Embedding(nc, hd)
        self.num_denoising, self.label_noise_ratio, self.box_noise_scale = nd, label_noise_ratio, box_noise_scale
        self.learnt_init_query = learnt_init_query
        self.query_pos_head = MLP(4, 2 * hd, hd, num_layers=2)
        self.enc_output = nn.Sequential(nn.Linear(hd, hd), nn.LayerNorm(hd))
        self.enc_score_head = nn.Linear(hd, nc)
        self.enc_bbox_head = MLP(hd, hd, 4, num_layers=3)
        self.dec_score_head = nn.ModuleList([nn.Linear(hd, nc) for _ in range(ndl)])
        self.dec_bbox_head = nn.ModuleList([MLP(hd, hd, 4, num_layers=3) for _ in range(ndl)])
        self._reset_parameters()

    def _reset_parameters(self):
        """The reference's initialisation recipe, head.py:2202-2224, restated as is: a freshly built decoder has to hold the same
        parameter values as the reference's (same RNG draws in the same order, same constants) for state_dict-level parity, so the
        sequence of init calls is fixed by the reference and not a design choice of this repository."""
        bias_cls = float(-math.log((1 - 0.01) / 0.01)) / 80 * self.nc
        nn.init.constant_(self.enc_score_head.bias, bias_cls)
        nn.init.constant_(self.enc_bbox_head.layers[-1].weight, 0.0)
        nn.init.constant_(self.enc_bbox_head.layers[-1].bias, 0.0)
        for cls_, reg_ in zip(self.dec_score_head, self.dec_bbox_head):
            nn.init.constant_(cls_.bias, bias_cls)
            nn.init.constant_(reg_.layers[-1].weight, 0.0)
            nn.init.constant_(reg_.layers[-1].bias, 0.0)
        nn.init.xavier_uniform_(self.enc_output[0].weight)
        nn.init.xavier_uniform_(self.query_pos_head.layers[0].weight)
        nn.init.xavier_uniform_(self.query_pos_head.layers[1].weight)
        for layer in self.input_proj:
            nn.init.xavier_uniform_(layer[0].weight)

    @staticmethod
    def _generate_anchors(shapes, grid_size=0.05, eps=1e-2):
        """Logit-space anchors + validity mask for all tokens: the constant table of head.py:2078-2115, restated operation for
        operation because the f32 values must be bit-identical to the reference's (they are added to the box logits before the
        top-k).  Tiny and input-independent, so it is computed once on the host and uploaded (`_static`)."""
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

    def _static(self, shapes, bs, device):
        """Per-(shapes, batch) constants: anchors (T,4), per-row valid mask in level-major order, level table."""
        cache = self.__dict__.setdefault("_static_cache", {})
        key = (tuple(map(tuple, shapes)), bs, str(device))
        st = cache.get(key)
        if st is None:
            anchors, valid = self._generate_anchors(shapes)
            v = valid.view(-1).float()
            rows, t0 = [], 0
            for h, w in shapes:  # level-major rows: every image repeats the level's mask
                rows.append(v[t0: t0 + h * w].repeat(bs))
                t0 += h * w
            hw = torch.tensor([h * w for h, w in shapes], dtype=torch.int32)
            shp = torch.tensor([d for s in shapes for d in s], dtype=torch.int32)
            st = dict(anchors=anchors.view(-1, 4).contiguous().to(device), rowmask=torch.cat(rows).contiguous().to(device),
                      hw_host=hw, shapes_host=shp, host_ptr=shp.data_ptr(), hw_ptr=hw.data_ptr(), T=int(hw.sum()))
            cache[key] = st
        return st

    def forward(self, x, batch=None):
        if self.training:
            raise L.UpaError("training-mode RTDETRDecoder is outside the hot-path scope (SURVEY §2 row 16)")
        lib = L.lib()
        x = [R.to_nhwc(t, t.dtype) for t in x]
        dev = x[0].device
        st_ = L.current_stream(dev)
        bs = x[0].shape[0]
        shapes = [[int(t.shape[2]), int(t.shape[3])] for t in x]
        st = self._static(shapes, bs, dev)
        hd, T = self.hidden_dim, st["T"]
        # ---- _get_encoder_input (head.py:2117-2141): 1x1 conv (BN folded) per level into the level-major token matrix
        feats = _Rows.new(bs * T, hd, dev, key=(id(self), "feats"))
        row0 = 0
        perf = any(t.dtype == torch.bfloat16 for t in x)
        fb = None
        if perf and self.proj_bf16 and all(t.dtype == torch.bfloat16 for t in x) and hd % 8 == 0:
            # perf mode: the three input projections (1x1 conv + folded BN, head.py:2117-2141) run as bf16 convs on the bf16 feature maps
            # and write the bf16 token matrix the value projections read; the float32 copy the ranking path works on is ONE widening
            # pass over it (before: three widening passes over the feature maps + three exact-f32 convs, 3 x (20 + 130) us at bs 16)
            fb = R.alloc_nhwc(1, hd, 1, bs * T, torch.bfloat16, dev, key=(id(self), "feats_bf16"))
            vfb = R.view_of(fb)
            for i, t in enumerate(x):
                conv, bn = self.input_proj[i][0], self.input_proj[i][1]
                cache = self.__dict__.setdefault("_pk_cache", {})
                ver = version_key(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
                hit = cache.get(("proj16", i, str(dev)))
                if hit is not None and hit[0] == ver:
                    pk = hit[1]
                else:
                    w, b = fold_bn(conv, bn)
                    pk = PackedConv(w, b, 1, dev, torch.bfloat16, False)
                    cache[("proj16", i, str(dev))] = (ver, pk)
                v = R.view_of(t)
                L.check(lib.upa_conv2d_bias_act(v.ptr, v.n, v.h, v.w, v.c, v.ld, pk.w.data_ptr(), pk.bias.data_ptr(),
                                                vfb.ptr + row0 * vfb.ld * 2, hd, vfb.ld, None, 0, 1, 1, 0, L.ACT_NONE,
                                                L.UPA_BF16, R.opts_ptr(), st_), "input_proj")
                row0 += bs * v.h * v.w
            L.check(lib.upa_cast_view(vfb.ptr, L.UPA_BF16, vfb.ld, feats.data_ptr(), L.UPA_F32, hd, bs * T, hd, st_), "widen_tokens")
            x = []
        for i, t in enumerate(x):
            if t.dtype == torch.bfloat16:
                # bf16 backbone with `proj_bf16` off: the decoder input stays float32 - the three feature maps are widened once here
                vt = R.view_of(t)
                tf = R.alloc_nhwc(vt.n, vt.c, vt.h, vt.w, torch.float32, dev, key=(id(self), "widen", i))
                vf = R.view_of(tf)
                L.check(lib.upa_cast_view(vt.ptr, L.UPA_BF16, vt.ld, vf.ptr, L.UPA_F32, vf.ld, vt.n * vt.h * vt.w, vt.c, st_),
                        "cast_view")
                t = tf
            elif t.dtype != torch.float32:
                raise L.UpaError(f"RTDETRDecoder takes float32 or bfloat16 feature maps, got {t.dtype}")
            conv, bn = self.input_proj[i][0], self.input_proj[i][1]
            cache = self.__dict__.setdefault("_pk_cache", {})
            ver = version_key(conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
            hit = cache.get(("proj", i, str(dev)))
            if hit is not None and hit[0] == ver:
                pk = hit[1]
            else:
                w, b = fold_bn(conv, bn)
                pk = PackedConv(w, b, 1, dev, torch.float32, False)
                cache[("proj", i, str(dev))] = (ver, pk)
            v = R.view_of(t)
            L.check(lib.upa_conv2d_bias_act(v.ptr, v.n, v.h, v.w, v.c, v.ld, pk.w.data_ptr(), pk.bias.data_ptr(),
                                            feats.data_ptr() + row0 * hd * 4, hd, hd, None, 0, 1, 1, 0, L.ACT_NONE,
                                            L.UPA_F32, R.opts_ptr(), st_), "input_proj")
            row0 += bs * v.h * v.w
        # ---- _get_decoder_input (head.py:2143-2200)
        saved_lin = _LINEAR_BF16[0]
        _LINEAR_BF16[0] = bool(perf and self.linear_bf16)
        try:
            return self._decode(feats, st, bs, hd, T, perf, dev, st_, lib, fb)
        finally:
            _LINEAR_BF16[0] = saved_lin

    # Parity instrumentation (tests / bench parity legs only; both None in production).  `query_override`: a (bs, num_queries) integer
    # tensor of reference token indices (index into the concatenated levels of ONE image, the `topk_ind` of head.py:2175) that replaces
    # the query selection - the decoder then refines exactly the tokens a reference run selected, so its output can be compared row by
    # row although the selection itself is chaotic in reduced precision (the top-300 of nearly flat scores).  `taps`: a dict that
    # receives the encoder-side tensors of the call (`features` = enc_output rows, `enc_scores` = enc_score_head logits, both
    # level-major, see `level_major_to_image`).
    query_override = None
    taps = None

    def token_rows(self, tok, st, bs):
        """(bs, nq) reference token indices -> (level-major row index, token index) int32 vectors of length bs * nq, the two outputs
        of upa_topk_tokens (include/upa.h): row = sum_{l' < l} bs * hw[l'] + image * hw[l] + p for token = sum_{l' < l} hw[l'] + p."""
        tok = torch.as_tensor(tok).to("cpu", torch.int64)
        if tok.dim() != 2 or tok.shape[0] != bs or tok.shape[1] != self.num_queries:
            raise L.UpaError(f"query_override must be ({bs}, {self.num_queries}), got {tuple(tok.shape)}")
        hw = st["hw_host"].to(torch.int64)
        if int(tok.min()) < 0 or int(tok.max()) >= int(hw.sum()):
            raise L.UpaError("query_override holds a token index outside the encoder's token range")
        start = torch.cumsum(hw, 0) - hw                       # first token of every level inside one image
        lvl = torch.bucketize(tok, start, right=True) - 1
        p = tok - start[lvl]
        img = torch.arange(bs, dtype=torch.int64).unsqueeze(1)
        rows = bs * start[lvl] + img * hw[lvl] + p
        return rows.reshape(-1).to(torch.int32), tok.reshape(-1).to(torch.int32)

    @staticmethod
    def level_major_to_image(rows, st, bs):
        """A level-major token matrix (bs * T, c) as the reference's (bs, T, c) tensor (head.py:2139: levels concatenated per image)."""
        out, r0 = [], 0
        for n in st["hw_host"].tolist():
            out.append(rows[r0: r0 + bs * n].view(bs, n, -1))
            r0 += bs * n
        return torch.cat(out, 1)

    # perf mode (bf16 backbone): the decoder's nn.Linear products on the bf16 matrix cores (see `_LINEAR_BF16`); False = exact float32
    linear_bf16 = True
    # perf mode: the input projections as bf16 convs straight into the bf16 token matrix; False = widen + exact-f32 convs
    proj_bf16 = True

    def _decode(self, feats, st, bs, hd, T, perf, dev, st_, lib, fb=None):
        masked = _Rows.new(bs * T, hd, dev, key=(id(self), "masked"))
        L.check(lib.upa_rows_scale(feats.data_ptr(), st["rowmask"].data_ptr(), masked.data_ptr(), bs * T, hd, st_), "mask")
        eo = self.enc_output
        features = layer_norm(eo[1], linear(self, "enc_output", eo[0].weight, eo[0].bias, masked, key=(id(self), "eo")),
                              key=(id(self), "eo_ln"))
        scores = linear(self, "enc_score", self.enc_score_head.weight, self.enc_score_head.bias, features,
                        key=(id(self), "enc_scores"))
        nq = self.num_queries
        rows = R.alloc_plain((bs * nq,), torch.int32, dev, key=(id(self), "topk_rows"))
        toks = R.alloc_plain((bs * nq,), torch.int32, dev, key=(id(self), "topk_tok"))
        if self.taps is not None:
            self.taps.update(features=features, enc_scores=scores, static=st, bs=bs)
        if self.query_override is None:
            L.check(lib.upa_topk_tokens(scores.data_ptr(), self.nc, self.nl, st["hw_ptr"], bs, nq, rows.data_ptr(),
                                        toks.data_ptr(), st_), "topk_tokens")
        else:
            r_, t_ = self.token_rows(self.query_override, st, bs)
            rows.copy_(r_.to(dev))
            toks.copy_(t_.to(dev))
        top_feat = _Rows.new(bs * nq, hd, dev, key=(id(self), "top_feat"))
        L.check(lib.upa_rows_gather(features.data_ptr(), rows.data_ptr(), top_feat.data_ptr(), bs * nq, hd, st_), "gather")
        delta = self.enc_bbox_head(top_feat, key=(id(self), "enc_bbox"))
        refer_logit = _Rows.new(bs * nq, 4, dev, key=(id(self), "refer_logit"))
        L.check(lib.upa_box_add_anchors(delta.data_ptr(), toks.data_ptr(), st["anchors"].data_ptr(), refer_logit.data_ptr(),
                                        bs * nq, st_), "add_anchors")
        # ---- decoder (transformer.py:719-773)
        projected = None
        if perf and self.fuse_value_proj:
            # perf mode (bf16 backbone): the value projection of every decoder layer reads the SAME encoder tokens
            # (transformer.py:536, value = feats), so the ndl projections are one (rows, ndl * C) x C GEMM on bf16 MFMA
            # instead of ndl exact-f32 GEMMs of 134400 x 256 x 256 (6 x 254 us at bs 16); the deformable sampling reads the
            # bf16 rows with the layer's column offset.  Everything that ranks or refines queries stays float32.
            nl_ = self.num_decoder_layers if self.decoder.eval_idx < 0 else self.decoder.eval_idx + 1
            if fb is None:
                fb = R.alloc_nhwc(1, hd, 1, bs * T, torch.bfloat16, dev, key=(id(self), "feats_bf16"))
                vf = R.view_of(fb)
                L.check(lib.upa_cast_view(feats.data_ptr(), L.UPA_F32, hd, vf.ptr, L.UPA_BF16, vf.ld, bs * T, hd, st_), "cast_feats")
            cache = self.__dict__.setdefault("_pk_cache", {})
            ws = [self.decoder.layers[i].cross_attn.value_proj for i in range(nl_)]
            ver = version_key(*[t for m_ in ws for t in (m_.weight, m_.bias)])
            hit = cache.get(("value_all", str(dev)))
            if hit is not None and hit[0] == ver:
                pk = hit[1]
            else:
                w = torch.cat([m_.weight.detach().float().cpu() for m_ in ws], 0).view(nl_ * hd, hd, 1, 1)
                b = torch.cat([m_.bias.detach().float().cpu() for m_ in ws], 0)
                pk = PackedConv(w, b, 1, dev, torch.bfloat16, False)
                cache[("value_all", str(dev))] = (ver, pk)
            va = R.alloc_nhwc(1, nl_ * hd, 1, bs * T, torch.bfloat16, dev, key=(id(self), "value_all"))
            hip_conv2d(fb, pk, 1, 0, L.ACT_NONE, out=va)
            vv = R.view_of(va)
            projected = (vv.ptr, L.UPA_BF16, vv.ld)
        boxes, dec_scores = self.decoder(top_feat, refer_logit, feats, st, bs, self.dec_bbox_head, self.dec_score_head,
                                         self.query_pos_head, projected=projected)
        y = R.alloc_plain((bs, nq, 4 + self.nc), torch.float32, dev, key=(id(self), "y"))
        L.check(lib.upa_rtdetr_output(boxes.data_ptr(), dec_scores.data_ptr(), y.data_ptr(), bs * nq, self.nc, st_),
                "rtdetr_output")
        return y if self.export else (y, (boxes.view(1, bs, nq, 4), dec_scores.view(1, bs, nq, self.nc), None, None, None))

    def train(self, mode: bool = True):
        self.__dict__.pop("_pk_cache", None)
        return super().train(mode)
