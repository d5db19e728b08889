"""``MaskFormer3D`` for SAPIEN objects (reference: models/segnet_sapien.py:11-52):
two SA levels (N/2 MSG r={0.1,0.2}, N/4 r=0.4; 64 neighbours) and two FP levels."""
from ._segnet import BN_CONFIG, MaskFormer3DBase


class MaskFormer3D(MaskFormer3DBase):
    def __init__(self, n_slot, n_point=512, use_xyz=True, bn=BN_CONFIG, n_transformer_layer=2,
                 transformer_embed_dim=256, transformer_input_pos_enc=False):
        sa = [dict(div=2, radii=[0.1, 0.2], nsamples=[64, 64], mlps=[[3, 64, 64, 64], [3, 64, 64, 128]]),
              dict(div=4, radius=0.4, nsample=64, mlp=[64 + 128, 128, 128, 256])]
        fp = [[128 + 3, 128, 128, 64], [256 + 64 + 128, 256, 128]]
        super().__init__(sa, fp, n_slot, n_point, use_xyz, bn, n_transformer_layer, transformer_embed_dim,
                         transformer_input_pos_enc)
