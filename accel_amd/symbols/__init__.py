from . import accel_18
from . import accel_34
from . import accel_50
from . import accel_101
from . import accel_dff
from . import resnet_v1_101_deeplab_dcn
