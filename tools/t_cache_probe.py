"""Does the contraction read T faster when T fits the 256 MB Infinity Cache?  
C3's tensor with so few models that T = columns x 300 x 304 x 8 B stays below the cache size, against the full
C3 width: prints the contraction's GB/s and the TTM's TFLOP/s for each width (plan M, no line search)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CALS_HIP_TREE", "M")
import cp_cals_amd as cc
from cp_cals_amd import inputs

modes = [300, 300, 300]
X = inputs.tensor(modes, 0)
for n_models in (12, 24, 36, 64, 128, 256):
    ranks = inputs.ranks_1_to_20(n_models)
    R = sum(ranks)
    e = cc.Engine(modes, R)
    e.set_tensor(X)
    e.set_params(cc.default_params(max_iterations=10 ** 9, force_max_iter=1))
    for fs, lam in inputs.model_factors(modes, ranks, 1):
        e.enqueue(cc.Model(fs, lam))
    e.admit()
    e.sweep(4)
    e.synchronize()
    e.set_profiling(2)
    e.reset_kernel_stats()
    e.sweep(20)
    e.synchronize()
    ks = e.kernel_stats()
    t_mb = ((R + 127) // 128 * 128) * 300 * 304 * 8 / 1e6
    print("models %4d  columns %5d  T %7.1f MB | contraction %6.1f us, %6.0f GB/s | TTM %8.1f us, %5.1f TFLOP/s" % (
        n_models, R, t_mb, ks.contract_ms / max(ks.contract_launches, 1) * 1e3,
        ks.contract_bytes / (ks.contract_ms * 1e-3) * 1e-9 if ks.contract_ms else 0.0,
        ks.ttm_ms / max(ks.ttm_launches, 1) * 1e3, ks.ttm_flops / (ks.ttm_ms * 1e-3) * 1e-12 if ks.ttm_ms else 0.0))
    e.close()
