#!/usr/bin/env python3
"""Prints the kernel times, the sustained MFMA figure and the value / median / blocks of the last bench line in gpurun_out/bench.log."""
import json,sys
l=[x for x in open("gpurun_out/bench.log") if x.startswith("{")][-1]
d=json.loads(l)
print(d["kernels_ms"], d["sustained_mfma_peak"]["value"], d["value"], d["value_median"], d["ms_per_step_blocks"])
