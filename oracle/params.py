"""Closed-form deterministic parameters keyed by state_dict name.

TEST INFRASTRUCTURE ONLY.  Lets the golden fixtures store only inputs and
outputs: both the imported reference (oracle/gen_golden.py) and the tests
rebuild identical weights from the parameter *names*.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def det_tensor(name: str, shape, dtype=torch.float32, salt: int = 0) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) + 7919 * salt))
    shape = tuple(shape)
    if len(shape) == 4 and shape[1] == 1 and "sparseModel" in name:   # SparseConvNet's (volume, 1, nIn, nOut) checkpoint layout
        return det_tensor(name, (shape[0], shape[2], shape[3]), dtype, salt).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        a = rng.uniform(0.5, 1.5, shape)
    elif leaf == "running_mean":
        a = rng.normal(0.0, 0.1, shape)
    elif leaf == "weight" and len(shape) == 1:  # BN gamma
        a = rng.uniform(0.6, 1.4, shape)
    elif leaf == "bias":
        a = rng.normal(0.0, 0.05, shape)
    else:  # conv / linear / sparse-conv weights: He-like fan-in scaling
        if len(shape) == 4:      # (Cout, Cin, kh, kw)  (ConvT: (Cin, Cout, 2, 2) -- close enough)
            fan = shape[1] * shape[2] * shape[3]
        elif len(shape) == 3:    # sparse conv (K, Cin, Cout)
            fan = shape[0] * shape[1]
        else:
            fan = shape[-1]
        a = rng.normal(0.0, np.sqrt(2.0 / fan), shape)
    return torch.from_numpy(np.asarray(a)).to(dtype)


def det_state(shapes: dict, dtype=torch.float32, salt: int = 0) -> dict:
    return {k: det_tensor(k, s, dtype, salt) for k, s in shapes.items()}
