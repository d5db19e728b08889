"""``FlowStep3D`` for SAPIEN (reference: models/flownet_sapien.py): width 128, 2-level global encoder
(N/8 k=16, N/16 k=8), 16/4 neighbours in the regressors / h0 net."""
from ._flownet import (GRU, EncoderGlob, EncoderLoc, Flow0Regressor, FlowRegressor, FlowStep3DBase, GlobalCorrLayer,
                       H0Net)

CONFIG = dict(
    width=128, reg_nsample=16, h0_nsample=4, loc_nsample=16, flow_conv_nsample=(8, 4),
    glob_enc=[(8, 16, 64, [128, 128, 128]), (16, 8, 128, [256, 256, 256])],
    glob_corr_sa=[(8, 8, 3, [32, 64, 128])],
)


class FlowStep3D(FlowStep3DBase):
    def __init__(self, npoint=512, use_instance_norm=False, loc_flow_nn=8, loc_flow_rad=0.1, k_decay_fact=1.0):
        super().__init__(CONFIG, npoint, use_instance_norm, loc_flow_nn, loc_flow_rad, k_decay_fact)
