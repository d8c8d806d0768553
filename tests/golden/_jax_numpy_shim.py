"""Numpy-backed stand-in for the tiny part of the `jax` namespace that the reference's
``timemachine/potentials/{nonbonded,bonded,jax_utils}.py`` and ``timemachine/integrator.py``
touch at import time and when evaluating *energies*; plus what ``timemachine/md/hrex.py`` (lax.scan / lax.cond in
_run_neighbor_swaps) and ``timemachine/md/barostat/moves.py`` (jax.ops.segment_sum in CentroidRescaler) need.

TEST INFRASTRUCTURE ONLY. jax/jaxlib are not installable in the build container (no network), so
the golden-vector generator (generate_golden.py) materialises this shim in a temporary directory
OUTSIDE the repo, puts it on sys.path in front of /root/reference, and then runs the reference's
own, unmodified Python source to produce energies.  jax.grad is *not* emulated: gradients in the
golden files come from oracle/ref_potentials.py (torch-f64 autograd), which the generator first
checks against these reference energies and against central finite differences of them.

This file contains no reference code.
"""
import os
import textwrap


def materialise(root: str) -> None:
    """Write the shim package tree under ``root`` (``root/jax/...``)."""
    files = {
        "jax/__init__.py": '''
            import numpy as _np
            from . import numpy, scipy, typing, core, random, ops, config as _cfgmod
            Array = _np.ndarray
            class _Config:
                def update(self, *a, **k):
                    pass
            config = _Config()
            def jit(f=None, **kw):
                if f is None:
                    return lambda g: g
                return f
            def pmap(f, *a, **k):
                return f
            def vmap(f, in_axes=0, out_axes=0):
                def g(*args):
                    n = None
                    if isinstance(in_axes, int):
                        axes = [in_axes] * len(args)
                    else:
                        axes = list(in_axes)
                    for a, ax in zip(args, axes):
                        if ax is not None:
                            n = _np.shape(a)[ax]
                            break
                    outs = []
                    for i in range(n):
                        sl = [(_np.take(a, i, axis=ax) if ax is not None else a) for a, ax in zip(args, axes)]
                        outs.append(f(*sl))
                    if isinstance(outs[0], tuple):
                        return tuple(_np.stack([o[k] for o in outs]) for k in range(len(outs[0])))
                    return _np.stack(outs)
                return g
            class custom_jvp:
                def __init__(self, f, nondiff_argnums=()):
                    self.f = f
                def __call__(self, *a, **k):
                    return self.f(*a, **k)
                def defjvp(self, g):
                    return g
            def grad(*a, **k):
                raise NotImplementedError("jax.grad is not available in the numpy shim")
            value_and_grad = grad
            class lax:
                @staticmethod
                def cond(pred, true_fun, false_fun, *operands):
                    return true_fun(*operands) if bool(pred) else false_fun(*operands)
                @staticmethod
                def scan(f, init, xs, length=None):
                    # xs: an array or a tuple of arrays scanned along axis 0; ys are stacked unless every one is None
                    carry = init
                    n = len(xs[0]) if isinstance(xs, tuple) else len(xs)
                    ys = []
                    for i in range(n):
                        x = tuple(a[i] for a in xs) if isinstance(xs, tuple) else xs[i]
                        carry, y = f(carry, x)
                        ys.append(y)
                    if all(y is None for y in ys):
                        return carry, None
                    return carry, _np.stack(ys)
                @staticmethod
                def fori_loop(lo, hi, body, val):
                    for i in range(lo, hi):
                        val = body(i, val)
                    return val
        ''',
        "jax/config.py": "",
        "jax/ops.py": '''
            import numpy as _np
            def segment_sum(data, segment_ids, num_segments=None):
                data = _np.asarray(data)
                segment_ids = _np.asarray(segment_ids)
                n = int(segment_ids.max()) + 1 if num_segments is None else num_segments
                out = _np.zeros((n,) + data.shape[1:], dtype=data.dtype)
                _np.add.at(out, segment_ids, data)
                return out
        ''',
        "jax/core.py": "class Tracer: pass\n",
        "jax/typing.py": "from typing import Any\nArrayLike = Any\n",
        "jax/random.py": "def PRNGKey(*a, **k):\n    raise NotImplementedError\n",
        "jax/numpy/__init__.py": '''
            import numpy as _np
            from numpy import linalg as _linalg
            class _At:
                def __init__(self, arr):
                    self.arr = arr
                def __getitem__(self, idx):
                    arr = self.arr
                    class _S:
                        def set(self_inner, v):
                            out = _np.array(arr, copy=True)
                            out[idx] = v
                            return out.view(_Arr)
                        def add(self_inner, v):
                            out = _np.array(arr, copy=True)
                            _np.add.at(out, idx, v)
                            return out.view(_Arr)
                    return _S()
            class _Arr(_np.ndarray):
                @property
                def at(self):
                    return _At(self)
            def _wrap_out(o):
                if isinstance(o, _np.ndarray) and not isinstance(o, _Arr):
                    return o.view(_Arr)
                if isinstance(o, tuple):
                    return tuple(_wrap_out(v) for v in o)
                return o
            def _wrap(f):
                def g(*a, **k):
                    return _wrap_out(f(*a, **k))
                g.__name__ = getattr(f, "__name__", "f")
                return g
            class _NS:
                pass
            linalg = _NS()
            for _n in dir(_linalg):
                _v = getattr(_linalg, _n)
                setattr(linalg, _n, _wrap(_v) if callable(_v) and not isinstance(_v, type) else _v)
            for _n in dir(_np):
                if _n.startswith("_") or _n == "linalg":
                    continue
                _v = getattr(_np, _n)
                if callable(_v) and not isinstance(_v, type):
                    globals()[_n] = _wrap(_v)
                else:
                    globals()[_n] = _v
        ''',
        "jax/scipy/__init__.py": "from . import special\n",
        "jax/scipy/special.py": "from scipy.special import erfc, logsumexp, erf  # noqa\n",
    }
    for rel, body in files.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            fh.write(textwrap.dedent(body))
