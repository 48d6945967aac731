#!/usr/bin/env python3
"""GPU box: cProfile of the host side of the training step at 540p (where the step is launch-bound)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--height", "540", "--width", "960", "--steps", "20", "--warmup", "3", "--cpu-baseline", "none"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
