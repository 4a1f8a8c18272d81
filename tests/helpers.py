"""Shared helpers of the parity tests (plain numpy; no reference code)."""
import numpy as np


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def make_models(inputs, modes, ranks, seed=1, jk=None):
    """[(factors, lam, jk)] with the jk fiber row zeroed, as Ktensor::fill does (ktensor.cpp:21-30)."""
    out = []
    for k, (fs, lam) in enumerate(inputs.model_factors(modes, ranks, seed)):
        j = None if jk is None else jk[k]
        if j is not None:
            fs[j[0]][j[1], :] *= 0.0
        out.append((fs, lam, j))
    return out


def numpy_mttkrp(X, modes, factors, mode):
    """Independent formulation of the MTTKRP (einsum), for cross-checking the oracle."""
    Xn = np.asarray(X).reshape(modes, order="F")
    letters = "ijklmnop"[: len(modes)]
    ins = [letters]
    ops = [Xn]
    for n, f in enumerate(factors):
        if n == mode:
            continue
        ins.append(letters[n] + "r")
        ops.append(f)
    return np.einsum(",".join(ins) + "->" + letters[mode] + "r", *ops)


def reconstruct(factors, lam, modes):
    letters = "ijklmnop"[: len(modes)]
    expr = "r," + ",".join(l + "r" for l in letters) + "->" + letters
    return np.einsum(expr, lam, *factors).ravel(order="F")
