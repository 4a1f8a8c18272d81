"""Diagnostics: C3 sweep time under the CALS_MTTKRP_* diagnostic switches (no clock stamps)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
e.sweep(10); e.synchronize()
e.set_profiling(True); e.reset_kernel_stats()
e.sweep(30); e.synchronize()
ks = e.kernel_stats()
print("%s: mttkrp %.4f ms/launch" % (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("CALS_")),
                                     ks.mttkrp_ms / ks.mttkrp_launches))
