from .ifnet import IFNet, IFNetFeatureExtractor, IFNetFeatureExtractor128, evaluate_network_on_grid, make_3d_grid  # noqa: F401
from .projection import project  # noqa: F401
from .unet import UNetMini, Unet  # noqa: F401
