from .pointnet2 import *  # noqa: F401,F403
