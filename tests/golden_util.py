"""Deterministic tensor fills and compact golden-vector storage shared by the fixture
generator (tests/golden/make_golden.py) and the tests.

Everything is keyed on strings and produced by numpy's PCG64, so the same arrays can be
regenerated here (next to the imported reference) and on the GPU box (next to the HIP path)
without shipping weights: a full ERFNet state_dict is 8.3 MB, far more than the outputs worth
pinning.
"""
import json
import os
import zlib

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_LIMIT = 8192      # arrays up to this many elements are stored whole
N_SAMPLES = 4096       # larger arrays: this many sampled elements + global statistics


def _rng(key):
    return np.random.default_rng(zlib.crc32(key.encode("utf-8")))


def det_normal(key, shape, scale=1.0):
    return (_rng(key).standard_normal(shape) * scale).astype(np.float32)


def det_uniform(key, shape, lo, hi):
    return _rng(key).uniform(lo, hi, shape).astype(np.float32)


def det_labels(key, shape, lo, hi):
    """Integer labels in [lo, hi) (int64)."""
    return _rng(key).integers(lo, hi, shape, dtype=np.int64)


def fill_value(key, shape, tag=""):
    """Closed-form deterministic value for one state_dict entry.

    Conv weights get variance 1/fan_in so that 70 stacked conv+ReLU layers keep O(1)
    activations while every residual branch still contributes visibly; BatchNorm statistics
    are kept strictly positive / away from degenerate values.
    """
    k = tag + "|" + key
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    parent = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
    is_bn = parent.startswith("bn")
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if is_bn:
        if leaf == "weight":
            return det_uniform(k, shape, 0.5, 1.5)
        if leaf == "bias":
            return det_uniform(k, shape, -0.2, 0.2)
        if leaf == "running_mean":
            return det_uniform(k, shape, -0.2, 0.2)
        if leaf == "running_var":
            return det_uniform(k, shape, 0.5, 1.5)
    if leaf == "weight" and len(shape) == 4:
        # Conv2d weight is [Cout, Cin, KH, KW]; for ConvTranspose2d ([Cin, Cout, KH, KW]) the
        # same expression only changes the scale by a small constant, which is harmless here.
        fan_in = shape[1] * shape[2] * shape[3]
        a = float(np.sqrt(3.0 / fan_in))
        return det_uniform(k, shape, -a, a)
    if leaf == "weight" and len(shape) == 2:
        a = float(np.sqrt(3.0 / shape[1]))
        return det_uniform(k, shape, -a, a)
    if leaf == "bias":
        return det_uniform(k, shape, -0.1, 0.1)
    if "class_power" in key:
        return det_uniform(k, shape, 0.5, 2.0)
    raise KeyError(f"no fill rule for state_dict key {key!r} shape {shape}")


def fill_state_dict(module, tag=""):
    """Overwrite every entry of ``module.state_dict()`` in place with ``fill_value``."""
    import torch
    sd = module.state_dict()
    with torch.no_grad():
        for key, t in sd.items():
            v = fill_value(key, t.shape, tag)
            t.copy_(torch.from_numpy(v).to(t.dtype))
    return module


# --------------------------------------------------------------------------- compact storage
def sample_index(name, size):
    return _rng("idx|" + name).integers(0, size, N_SAMPLES)


def pack(store, name, arr):
    """Record ``arr`` under ``name`` in dict ``store`` (whole, or samples + statistics)."""
    a = np.ascontiguousarray(np.asarray(arr))
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    if a.size <= FULL_LIMIT or a.dtype.kind in "iu":
        store[name] = a
        return
    flat = a.reshape(-1)
    store[name + "#samp"] = flat[sample_index(name, flat.size)]
    f64 = flat.astype(np.float64)
    store[name + "#stat"] = np.array([f64.sum(), np.abs(f64).sum(), np.abs(f64).max(),
                                      float(flat.size)], np.float64)
    store[name + "#shape"] = np.array(a.shape, np.int64)


def _within(err, ref, atol, rtol, scale_rtol, outlier_frac, outlier_rtol, label):
    """Elementwise |err| <= atol + rtol*|ref| + scale_rtol*max|ref|, except that a fraction
    ``outlier_frac`` of the elements may miss it as long as they stay within
    ``outlier_rtol*max|ref|``.

    The outlier allowance exists for gradients only: a ReLU whose pre-activation is within
    fp32 rounding of zero can switch its mask between two correct implementations, which
    perturbs the (otherwise linear) backward pass in a small neighbourhood.
    """
    if err.size == 0:
        return
    gmax = float(np.abs(ref).max())
    tol = atol + rtol * np.abs(ref) + scale_rtol * gmax
    bad = err > tol
    nbad = int(bad.sum())
    if nbad == 0:
        return
    assert outlier_frac > 0 and nbad <= outlier_frac * err.size and \
        float(err.max()) <= outlier_rtol * gmax + atol, \
        (f"{label}: {nbad}/{err.size} elements out of tolerance, max err {err.max():.3e} "
         f"(atol {atol}, rtol {rtol}, scale_rtol {scale_rtol}, max|ref| {gmax:.3e})")


def check(store, name, arr, atol, rtol=0.0, what="", scale_rtol=0.0, outlier_frac=0.0,
          outlier_rtol=0.0):
    """Assert ``arr`` matches the golden entry ``name``.  Returns the max abs error seen."""
    a = np.ascontiguousarray(np.asarray(arr))
    label = what or name
    if name in store:
        g = store[name]
        assert tuple(g.shape) == tuple(a.shape), f"{label}: shape {a.shape} vs golden {g.shape}"
        if g.dtype.kind in "iu":
            assert np.array_equal(g, a), f"{label}: integer mismatch"
            return 0.0
        g64 = g.astype(np.float64)
        err = np.abs(a.astype(np.float64) - g64)
        _within(err, g64, atol, rtol, scale_rtol, outlier_frac, outlier_rtol, label)
        return float(err.max()) if err.size else 0.0
    assert name + "#samp" in store, f"golden entry {name!r} missing"
    shape = tuple(int(s) for s in store[name + "#shape"])
    assert shape == tuple(a.shape), f"{label}: shape {a.shape} vs golden {shape}"
    flat = a.reshape(-1)
    g = store[name + "#samp"].astype(np.float64)
    got = flat[sample_index(name, flat.size)].astype(np.float64)
    err = np.abs(got - g)
    _within(err, g, atol, rtol, scale_rtol, outlier_frac, outlier_rtol, label + " (sampled)")
    gsum, gabs, gmax, gsize = store[name + "#stat"]
    f64 = flat.astype(np.float64)
    # mean-level statistics: error of a sum of n terms each within tol
    mean_tol = atol + rtol * gabs / gsize + scale_rtol * gmax + outlier_frac * outlier_rtol * gmax
    assert abs(f64.sum() - gsum) / gsize <= mean_tol, \
        f"{label}: mean differs {(f64.sum() - gsum) / gsize:.3e}"
    assert abs(np.abs(f64).sum() - gabs) / gsize <= mean_tol, f"{label}: abs-mean differs"
    assert abs(np.abs(f64).max() - gmax) <= atol + (rtol + scale_rtol + outlier_rtol + 1e-6) * gmax, \
        f"{label}: max differs"
    return float(err.max())


def check_arrays(got, ref, atol, rtol=0.0, what="", scale_rtol=0.0, outlier_frac=0.0,
                 outlier_rtol=0.0):
    """Same criterion as ``check`` for two in-memory arrays."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    r64 = ref.astype(np.float64)
    err = np.abs(got.astype(np.float64) - r64)
    _within(err, r64, atol, rtol, scale_rtol, outlier_frac, outlier_rtol, what)
    return float(err.max()) if err.size else 0.0


def load_npz(name):
    path = os.path.join(GOLDEN_DIR, name)
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def load_json(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)
