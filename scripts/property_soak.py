#!/usr/bin/env python3
"""The three hypothesis properties of the GPU suite with FRESH random examples (the suite's own runs are derandomised: the
same 150 / 40 examples every time): python3 scripts/property_soak.py [seed] [tiny-sensor examples] [sorted-cloud examples] [layout examples]"""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tests")); sys.path.insert(0, str(REPO / "point-cloud-preprocessing-tools_amd")); sys.path.insert(0, str(REPO))
from hypothesis import given, settings, HealthCheck, seed
import test_gpu_property as tp
import test_gpu_stream as ts

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n2 = int(sys.argv[3]) if len(sys.argv) > 3 else 400
n3 = int(sys.argv[4]) if len(sys.argv) > 4 else n2
for name, fn, strat, n in (("tiny sensors, arbitrary clouds", tp.test_tiny_sensors_match_oracle, tp.sensor_and_frames(), n1),
                           ("mid-sized sensors, sorted clouds with hidden defects", ts.test_sorted_clouds_of_mid_sized_sensors_match_oracle, ts._sorted_frames(), n2),
                           ("structured and firing-order layouts with hidden defects, row loop and tiles", tp.test_structured_and_firing_order_layouts_match_oracle, tp.layout_frames(), n3)):
    count = [0]
    def make_body(inner, count):
        def body(case):
            count[0] += 1
            if count[0] % 1000 == 0: print(f"  ... {count[0]} batches", flush=True)
            try:
                inner(case)
            except AssertionError:   # keep the drawn batch: hypothesis prints it abridged
                import pickle
                out = REPO / "gpurun_out" / "property_soak_failure.pkl"
                out.parent.mkdir(exist_ok=True)
                pickle.dump(case, open(out, "wb"))
                print("failing batch saved to", out, flush=True)
                raise
        return body
    body = make_body(fn.hypothesis.inner_test, count)
    t = settings(max_examples=n, deadline=None, suppress_health_check=list(HealthCheck), derandomize=False, database=None)(given(strat)(body))
    t0 = time.time()
    seed(s0)(t)()
    print(f"{name}: seed {s0}, {count[0]} drawn batches, all equal to the oracle, {time.time() - t0:.0f} s", flush=True)
