"""Deterministic, library-independent input generator (TEST INFRASTRUCTURE).

Golden fixtures under tests/golden/ store only OUTPUTS; inputs and parameters are regenerated
from (name, shape) by this counter-hash so that the fixture files stay small and the same bits
are produced here, on the GPU box and in every later round.  Nothing here depends on torch's
or numpy's RNG streams: splitmix64 over a per-tensor 64-bit key + element counter, top 53 bits
-> uniform in (0,1), Box-Muller -> normal.  All arithmetic is uint64/float64 and exact.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _key(name, seed):
    return np.uint64(zlib.crc32(name.encode()) | (int(seed) << 32))


def uniform01(name, shape, seed=0):
    """float64 uniform in (0, 1), fully determined by (name, shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + _splitmix64(np.array([_key(name, seed)], np.uint64))[0]
        bits = _splitmix64(ctr)
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    return u.reshape(shape)


def uniform(name, shape, lo=-1.0, hi=1.0, seed=0, dtype=np.float32):
    return (lo + (hi - lo) * uniform01(name, shape, seed)).astype(dtype)


def normal(name, shape, mean=0.0, std=1.0, seed=0, dtype=np.float32):
    u1 = uniform01(name + "/u1", shape, seed)
    u2 = uniform01(name + "/u2", shape, seed)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(dtype)


def randint(name, shape, lo, hi, seed=0):
    """int64 in [lo, hi)."""
    u = uniform01(name, shape, seed)
    return np.minimum((lo + np.floor(u * (hi - lo))).astype(np.int64), hi - 1)


def fill_state_dict(state_dict, seed=0, scale=None):
    """Deterministic parameters for any module: returns {key: float32 ndarray} with the shapes
    of `state_dict`.  Weights ~ U(+-1/sqrt(fan_in)) like the reference's Linear init
    (models/sit.py:50 -> torch default), LayerNorm gamma = 1 + 0.1 n, beta = 0.1 n (so that the
    affine terms are exercised), cls/pos/mask tokens ~ N(0,1) (models/sit.py:53-54)."""
    out = {}
    for k, v in state_dict.items():
        shp = tuple(v.shape)
        if k.endswith("norm.weight") or k.endswith("mlp_head.0.weight"):
            a = 1.0 + 0.1 * normal(k, shp, seed=seed)
        elif k.endswith("norm.bias") or k.endswith("mlp_head.0.bias"):
            a = 0.1 * normal(k, shp, seed=seed)
        elif k.endswith("pos_embedding") or k.endswith("cls_token") or k.endswith("mask_token"):
            a = normal(k, shp, seed=seed)
        elif k.endswith(".weight"):
            bound = 1.0 / np.sqrt(shp[-1])
            a = uniform(k, shp, -bound, bound, seed=seed)
        elif k.endswith(".bias"):
            a = uniform(k, shp, -0.05, 0.05, seed=seed)
        else:
            raise KeyError(k)
        out[k] = a.astype(np.float32)
    return out
