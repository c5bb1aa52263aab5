"""same-box A/B of Accel-101 with and without the split of its feature fusion (lower._plan_split_fusion):
    python scripts/ab_fusion101.py off|on [bench.py arguments]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
mode = sys.argv.pop(1)
if mode == "off":
    from accel_amd import lower
    lower.Lowering._plan_split_fusion = lambda self, cat: False
import bench
bench.main()
