"""MaskFormer-style transformer head on top of the point features (reference: utils/transformer_util.py).
torch.nn modules with the reference's parameter names; on the GPU the attention core between the projections is the
fused kernel of ogc_amd/csrc/attention.hip (ogc_amd.fused.multihead_attention), and nothing is created with the
reference's hard-coded ``.cuda()`` (transformer_util.py:110)."""
import torch
import torch.nn as nn


class TransformerDecoderLayer(nn.Module):
    """Cross-attention (slots <- points), self-attention (slots), feed-forward; pre-norm residuals.
    Reference: transformer_util.py:5-62."""

    def __init__(self, embed_dim=256, n_head=8, hidden_dim=256):
        super().__init__()
        self.norm_slot1 = nn.LayerNorm(embed_dim)
        self.norm_slot2 = nn.LayerNorm(embed_dim)
        self.norm_pre_ff = nn.LayerNorm(embed_dim)
        self.cross_attn = nn.MultiheadAttention(embed_dim, n_head, batch_first=True)
        self.self_attn = nn.MultiheadAttention(embed_dim, n_head, batch_first=True)
        self.mlp = nn.Sequential(nn.Linear(embed_dim, hidden_dim), nn.ReLU(inplace=True),
                                 nn.Linear(hidden_dim, embed_dim))

    def forward(self, slot, point_feats, pos_enc=None):
        # slot (B, K, C), point_feats (B, N, C), pos_enc (B, N, C) or None -> (B, K, C)
        from ..fused import multihead_attention  # the module's own parameters; fused attention core on the GPU
        if pos_enc is None:
            slot = slot + multihead_attention(self.cross_attn, self.norm_slot1(slot), point_feats, point_feats)
        else:
            slot = slot + self.cross_attn(query=self.norm_slot1(slot), key=point_feats + pos_enc, value=point_feats,
                                          need_weights=False)[0]
        s2 = self.norm_slot2(slot)
        slot = slot + multihead_attention(self.self_attn, s2, s2, s2)
        return slot + self.mlp(self.norm_pre_ff(slot))


class MaskFormerHead(nn.Module):
    """K learned queries decoded against the coarsest point features. Reference: transformer_util.py:65-121."""

    def __init__(self, n_slot, input_dim=256, n_transformer_layer=2, transformer_embed_dim=256,
                 transformer_n_head=8, transformer_hidden_dim=256, input_pos_enc=False):
        super().__init__()
        self.n_slot = n_slot
        self.query = nn.Embedding(n_slot, transformer_embed_dim)
        self.mlp_input = nn.Sequential(nn.Linear(input_dim, transformer_embed_dim), nn.ReLU(inplace=True),
                                       nn.Linear(transformer_embed_dim, transformer_embed_dim))
        self.norm_input = nn.LayerNorm(transformer_embed_dim)
        self.input_pos_enc = nn.Linear(3, transformer_embed_dim) if input_pos_enc else None
        self.transformer_layers = nn.ModuleList(
            TransformerDecoderLayer(embed_dim=transformer_embed_dim, n_head=transformer_n_head,
                                    hidden_dim=transformer_hidden_dim) for _ in range(n_transformer_layer))

    def forward(self, point_feats, point_pos):
        # point_feats (B, N, C_in), point_pos (B, N, 3) -> slots (B, K, D)
        n_batch = point_feats.shape[0]
        # every sample looks up slots 0..K-1 (transformer_util.py:108-111): the embedding table itself, broadcast —
        # same values, and the backward is one sum over the batch instead of an index sort + scatter
        slot = self.query.weight.unsqueeze(0).expand(n_batch, -1, -1)
        inputs = self.norm_input(self.mlp_input(point_feats))
        pos_enc = self.input_pos_enc(point_pos) if self.input_pos_enc is not None else None
        for layer in self.transformer_layers:
            slot = layer(slot, inputs, pos_enc)
        return slot
