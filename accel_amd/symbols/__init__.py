from . import accel_18
from . import accel_34
from . import accel_50
from . import accel_101
