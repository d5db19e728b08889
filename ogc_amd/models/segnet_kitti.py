"""``MaskFormer3D`` for KITTI-SF / Waymo scenes (reference: models/segnet_kitti.py:11-60):
three SA levels (N/4 MSG r={1,2}, N/8 r=4, N/16 r=8; 64 neighbours) and three FP levels."""
from ._segnet import BN_CONFIG, MaskFormer3DBase


class MaskFormer3D(MaskFormer3DBase):
    def __init__(self, n_slot, n_point=2048, use_xyz=True, bn=BN_CONFIG, n_transformer_layer=2,
                 transformer_embed_dim=256, transformer_input_pos_enc=False):
        sa = [dict(div=4, radii=[1, 2], nsamples=[64, 64], mlps=[[3, 32, 32, 32], [3, 32, 32, 64]]),
              dict(div=8, radius=4, nsample=64, mlp=[32 + 64, 64, 64, 128]),
              dict(div=16, radius=8, nsample=64, mlp=[128, 128, 128, 256])]
        fp = [[64 + 3, 64, 64, 64], [32 + 64 + 128, 64, 64], [128 + 256, 128, 128]]
        super().__init__(sa, fp, n_slot, n_point, use_xyz, bn, n_transformer_layer, transformer_embed_dim,
                         transformer_input_pos_enc)
