"""A-B timing of the echo kernel lane mappings: python tools/echo_ab.py 16|8|4 [channels] (prints ms per step, launch us)."""
import ctypes
import runpy
import sys

import torch  # noqa: F401  (before libspangpu: one HIP runtime must initialise first)

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from spandsp_amd import engine  # noqa: E402

engine.lib().spangpu_tune_echo_lanes_per_channel(int(sys.argv[1]))
chan = sys.argv[2:3]
sys.argv = ["bench_paths.py", "--workload", "echo", "--no-cpu-baseline", "--steps", "60"] + (["--channels", chan[0]] if chan else [])
runpy.run_path(__file__.rsplit("/", 1)[0] + "/bench_paths.py", run_name="__main__")
