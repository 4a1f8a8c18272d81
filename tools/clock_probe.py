"""Diagnostics: shader clock the MTTKRP kernel holds under load (needs CALS_MTTKRP_CLOCK=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CALS_MTTKRP_CLOCK"] = "1"
import cp_cals_amd as cc
from cp_cals_amd import inputs
modes = [300, 300, 300]
ranks = inputs.ranks_1_to_20(256)
X = inputs.tensor(modes, 0)
e = cc.Engine(modes, sum(ranks))
e.set_tensor(X)
e.set_params(cc.default_params(max_iterations=10**9, force_max_iter=1))
for fs, lam in inputs.model_factors(modes, ranks, 1):
    e.enqueue(cc.Model(fs, lam))
e.admit()
e.sweep(20); e.synchronize()
t = time.time(); e.sweep(40); e.synchronize(); dt = time.time() - t
cyc, ghz = e.debug_clock(252)
print("kernel=%s: %.3f ms/sweep; median workgroup: %.0f shader cycles, clock %.3f GHz" % (
    "v3", dt / 40 * 1e3, cyc, ghz))
