"""accel_amd: MI355X-native Accel (dff_deeplab) video-segmentation inference path."""
__version__ = "0.1.0"
