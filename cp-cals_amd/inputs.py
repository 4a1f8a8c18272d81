"""Portable synthetic inputs for the CALS hot path.

A counter-based generator (splitmix64) so that numpy here, C in the oracle and any other host
language produce bit-identical doubles: element i of stream `seed` is
    z = seed + (i+1) * 0x9E3779B97F4A7C15  (mod 2^64), mixed by the splitmix64 finaliser,
    u = (z >> 11) * 2^-53  in [0,1),  value = 2u - 1  in [-1,1).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def uniform_pm1(seed, n, offset=0):
    """n doubles U[-1,1) of stream `seed`, starting at element `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return 2.0 * u - 1.0


def tensor(modes, seed=0):
    """Dense X, linear index i0 + I0*i1 + I0*I1*i2 ... (mode 0 fastest), flat float64 array."""
    return uniform_pm1(seed, int(np.prod(modes)))


def ranks_1_to_20(n_models):
    """r_m = 1 + (m mod 20)  (SURVEY.md section 8d)."""
    return [1 + (m % 20) for m in range(n_models)]


def model_factors(modes, ranks, seed=1):
    """Per model: list of col-major (I_n x r) factors drawn in order (model, mode, col-major
    element) from stream `seed`, then per-column 2-norm normalisation with lambda = product of
    the norms (Ktensor::fill -> normalize(), src/ktensor.cpp:21-30,85-99)."""
    out = []
    off = 0
    for r in ranks:
        fs = []
        lam = np.ones(r)
        for I in modes:
            v = uniform_pm1(seed, I * r, off)
            off += I * r
            f = np.asfortranarray(v.reshape((I, r), order="F"))
            nrm = np.sqrt((f * f).sum(axis=0))
            f /= nrm
            lam *= nrm
            fs.append(f)
        out.append((fs, lam))
    return out


def low_rank_tensor(modes, rank, seed=7):
    """X = to_tensor(random rank-`rank` Ktensor) as the reference's Tensor(rank, modes) ctor does
    (src/tensor.cpp:81-87); returns (flat X, factors, lambda)."""
    (fs, lam), = model_factors(modes, [rank], seed)
    x = np.zeros(tuple(modes), order="F")
    if len(modes) == 3:
        x = np.einsum("r,ir,jr,kr->ijk", lam, fs[0], fs[1], fs[2])
    elif len(modes) == 4:
        x = np.einsum("r,ir,jr,kr,lr->ijkl", lam, fs[0], fs[1], fs[2], fs[3])
    else:
        raise ValueError("3- or 4-way only")
    return np.ascontiguousarray(x.ravel(order="F")), fs, lam
