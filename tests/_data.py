"""Seeded synthetic input classes shared by the CPU and GPU tests."""
import numpy as np


def bf16_round(a):
    """Round-to-nearest-even to bf16, returned widened to float32 (identical values on both sides)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def gaussian(n, d, seed):
    return np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)


def grid(n, d, seed):
    """integers in [-8, 8] * 2^-3: exactly bf16, every fp32 partial sum exact -> order independent."""
    return (np.random.default_rng(seed).integers(-8, 9, (n, d)) / 8.0).astype(np.float32)


def reaction_fp_like(n, d, seed, density=0.02):
    """sparse signed integer counts in [-10, 10] (RDKit difference fingerprint class,
    retrieve/retrieve_faiss.py:18-27)."""
    rng = np.random.default_rng(seed)
    mask = rng.random((n, d)) < density
    vals = rng.integers(-10, 11, (n, d))
    return (mask * vals).astype(np.float32)


def morgan_like(n, d, seed, density=0.05):
    """Bernoulli bit vectors (Morgan r=2 class, retrieve/retrieve_faiss.py:36-44)."""
    return (np.random.default_rng(seed).random((n, d)) < density).astype(np.float32)
