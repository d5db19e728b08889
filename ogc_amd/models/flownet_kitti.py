"""``FlowStep3D`` for KITTI-SF (reference: models/flownet_kitti.py): width 128, 3-level global encoder
(N/8 k=32, N/16 k=24, N/32 k=16), 32/8 neighbours in the regressors / h0 net."""
from ._flownet import (GRU, EncoderGlob, EncoderLoc, Flow0Regressor, FlowRegressor, FlowStep3DBase, GlobalCorrLayer,
                       H0Net, NoGRU)

CONFIG = dict(
    width=128, reg_nsample=32, h0_nsample=8, loc_nsample=32, flow_conv_nsample=(16, 8),
    glob_enc=[(8, 32, 64, [128, 128, 128]), (16, 24, 128, [128, 128, 128]), (32, 16, 128, [256, 256, 256])],
    glob_corr_sa=[(16, 16, 3, [32, 32, 64]), (8, 16, 64, [64, 64, 128])],
)


class FlowStep3D(FlowStep3DBase):
    def __init__(self, npoint=2048, use_instance_norm=False, loc_flow_nn=8, loc_flow_rad=0.1, k_decay_fact=1.0):
        super().__init__(CONFIG, npoint, use_instance_norm, loc_flow_nn, loc_flow_rad, k_decay_fact)
