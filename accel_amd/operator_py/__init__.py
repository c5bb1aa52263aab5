from .flow_warp import *   # registers 'FlowWarp'
from .tile_as import *     # registers 'tile_as'
