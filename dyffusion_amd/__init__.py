"""dyffusion_amd: MI355X-native DYffusion sampling engine behind the reference's module API.

Product path = HIP kernels in lib/libdyffusion_hip.so (include/dyffusion_hip.h); importing the modules that run the
hot path fails loudly when that library is missing.
"""
from .dyffusion import DYffusion  # noqa: F401
from .engine import EngineError, HipEngine, default_dtype_for, net_config, resnet_net_config  # noqa: F401
from .experiment import InterpolationExperiment, InterpolatorHandle, MultiHorizonForecastingDYffusion  # noqa: F401
from . import checkpoint  # noqa: F401
from . import metrics  # noqa: F401
from .unet import Unet  # noqa: F401
from .unet_simple import UNet  # noqa: F401
from .simple_conv_net import SimpleConvNet  # noqa: F401
from .boundary import PhysicalSystemsBoundaryConditions  # noqa: F401
