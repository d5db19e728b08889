"""MaskFormer-style transformer head on top of the point features (reference: utils/transformer_util.py).
torch.nn modules with the reference's parameter names (checkpoints are interchangeable); on the GPU the attention core
between the projections is the fused kernel of ogc_amd/csrc/attention.hip (ogc_amd.fused.multihead_attention), and
nothing is created with the reference's hard-coded ``.cuda()`` (transformer_util.py:110)."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import fused


def _attend(mha, query, key, value):
    """Output of a batch-first nn.MultiheadAttention; the fused core when key and value are the same tensor."""
    if key is value:
        from ..fused import multihead_attention
        return multihead_attention(mha, query, key, value)
    return mha(query=query, key=key, value=value, need_weights=False)[0]


class TransformerDecoderLayer(nn.Module):
    """Pre-norm residual block of three steps — slots attend to the points, slots attend to each other, feed-forward
    (reference: transformer_util.py:5-62)."""

    def __init__(self, embed_dim=256, n_head=8, hidden_dim=256):
        super().__init__()
        for name in ("norm_slot1", "norm_slot2", "norm_pre_ff"):
            setattr(self, name, nn.LayerNorm(embed_dim))
        for name in ("cross_attn", "self_attn"):
            setattr(self, name, nn.MultiheadAttention(embed_dim, n_head, batch_first=True))
        ff = [nn.Linear(embed_dim, hidden_dim), nn.ReLU(inplace=True), nn.Linear(hidden_dim, embed_dim)]
        self.mlp = nn.Sequential(*ff)

    def forward(self, slot, point_feats, pos_enc=None):
        # slot (B, K, C), point_feats (B, N, C), pos_enc (B, N, C) or None -> (B, K, C)
        keys = point_feats if pos_enc is None else point_feats + pos_enc
        slot = slot + _attend(self.cross_attn, self.norm_slot1(slot), keys, point_feats)
        inner = self.norm_slot2(slot)
        slot = slot + _attend(self.self_attn, inner, inner, inner)
        lin0, _, lin1 = self.mlp  # Linear, ReLU, Linear on B * K rows: one launch per layer and direction on the GPU
        hidden = F.relu(fused.small_linear(self.norm_pre_ff(slot), lin0.weight, lin0.bias))
        return slot + fused.small_linear(hidden, lin1.weight, lin1.bias)


class MaskFormerHead(nn.Module):
    """K learned queries decoded against the coarsest point features (reference: transformer_util.py:65-121)."""

    def __init__(self, n_slot, input_dim=256, n_transformer_layer=2, transformer_embed_dim=256,
                 transformer_n_head=8, transformer_hidden_dim=256, input_pos_enc=False):
        super().__init__()
        width = transformer_embed_dim
        self.n_slot = n_slot
        self.query = nn.Embedding(n_slot, width)
        self.mlp_input = nn.Sequential(nn.Linear(input_dim, width), nn.ReLU(inplace=True), nn.Linear(width, width))
        self.norm_input = nn.LayerNorm(width)
        self.input_pos_enc = nn.Linear(3, width) if input_pos_enc else None
        layers = [TransformerDecoderLayer(embed_dim=width, n_head=transformer_n_head, hidden_dim=transformer_hidden_dim)
                  for _ in range(n_transformer_layer)]
        self.transformer_layers = nn.ModuleList(layers)

    def forward(self, point_feats, point_pos):
        # point_feats (B, N, C_in), point_pos (B, N, 3) -> slots (B, K, D)
        # every sample looks up slots 0..K-1 (transformer_util.py:108-111): the embedding table itself, broadcast —
        # same values, and the backward is one sum over the batch instead of an index sort + scatter
        slot = self.query.weight.unsqueeze(0).expand(point_feats.shape[0], -1, -1)
        lin0, _, lin1 = self.mlp_input  # Linear, ReLU, Linear on B * N rows: weight gradients split over the batch
        hidden = F.relu(fused.many_rows_linear(point_feats, lin0.weight, lin0.bias))
        memory = self.norm_input(fused.many_rows_linear(hidden, lin1.weight, lin1.bias))
        pos_enc = None if self.input_pos_enc is None else self.input_pos_enc(point_pos)
        for layer in self.transformer_layers:
            slot = layer(slot, memory, pos_enc)
        return slot
