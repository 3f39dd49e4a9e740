"""PCG64 seeding with the same construction gymnasium 0.29.1 documents:
Generator(PCG64(SeedSequence(seed)))."""
import numpy as np


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError(f"Seed must be a non-negative integer or None, got {seed!r}")
    seq = np.random.SeedSequence(seed)
    np_seed = seq.entropy
    return np.random.Generator(np.random.PCG64(seq)), np_seed
