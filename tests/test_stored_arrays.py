"""StoredArrays / FileClient / per-leg result files (timemachine_amd/stored_arrays.py).  Behaviours checked are the ones
the reference's tests/test_stored_arrays.py pins: extend + iterate, integer (incl. negative) indexing, no references
held, equality, temp-dir cleanup, store/load and pickle round trips, refusal to overwrite."""

import gc
import pickle
import weakref
from pathlib import Path

import numpy as np
import pytest

from timemachine_amd.stored_arrays import (
    FileClient,
    StoredArrays,
    deserialize_array,
    load_leg_results,
    run_and_store_frames,
    save_leg_results,
    serialize_array,
)


def make_chunks(seed, n_chunks=4, shape=(5, 3), dtype=np.float64):
    rng = np.random.default_rng(seed)
    return [[rng.normal(size=shape).astype(dtype) for _ in range(rng.integers(0, 4))] for _ in range(n_chunks)]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_extend_iter_getitem(dtype):
    chunks = make_chunks(0, dtype=dtype)
    sa = StoredArrays.from_chunks(chunks)
    flat = [a for c in chunks for a in c]
    assert len(sa) == len(flat)
    for got, want in zip(sa, flat):
        np.testing.assert_array_equal(got, want)
        assert got.dtype == dtype
    for ix in range(-len(flat), len(flat)):
        np.testing.assert_array_equal(sa[ix], flat[ix])
    with pytest.raises(IndexError):
        sa[len(flat)]
    with pytest.raises(NotImplementedError, match="slices are not implemented"):
        sa[0:2]
    with pytest.raises(ValueError, match="invalid subscript"):
        sa["a"]


def test_holds_no_references_and_cleans_up():
    sa = StoredArrays()
    data = np.arange(10)
    sa.extend([data])
    ref = weakref.ref(data)
    del data
    gc.collect()
    assert ref() is None
    path = Path(sa._dir.name)
    assert path.exists()
    del sa
    gc.collect()
    assert not path.exists()


def test_eq_neq_pickle():
    chunks = make_chunks(1)
    a, b = StoredArrays.from_chunks(chunks), StoredArrays.from_chunks(chunks)
    assert a == a and a == b and a is not b
    c = StoredArrays.from_chunks(make_chunks(2))
    assert a != c
    # same arrays, different chunking -> different
    d = StoredArrays.from_chunks([[x for ch in chunks for x in ch]])
    assert a != d
    nan = StoredArrays.from_chunks([[np.array([np.nan, 1.0])]])
    assert nan == StoredArrays.from_chunks([[np.array([np.nan, 1.0])]])
    assert pickle.loads(pickle.dumps(a)) == a


def test_store_load_roundtrip_and_collision(tmp_path):
    fc = FileClient(tmp_path)
    sa = StoredArrays.from_chunks(make_chunks(3))
    sa.store(fc)
    assert StoredArrays.load(fc) == sa
    with pytest.raises(FileExistsError):
        sa.store(fc)
    sa.store(fc, prefix=Path("subdir"))
    assert StoredArrays.load(fc, prefix=Path("subdir")) == sa
    assert len(StoredArrays.load(fc, prefix=Path("nothing_here"))) == 0


def test_serialize_roundtrip_and_file_client(tmp_path):
    x = np.random.default_rng(0).normal(size=(4, 3))
    np.testing.assert_array_equal(deserialize_array(serialize_array(x)), x)
    fc = FileClient(tmp_path)
    fc.store("a/b.bin", b"hello")
    assert fc.exists("a/b.bin") and fc.load("a/b.bin") == b"hello"
    fc.delete("a/b.bin")
    assert not fc.exists("a/b.bin")


class FakeContext:
    """stands in for custom_ops.Context: returns the frames multiple_steps(n, interval) would"""

    def __init__(self, n_atoms):
        self.t = 0
        self.n = n_atoms

    def multiple_steps(self, n_steps, interval):
        n_frames = n_steps // interval
        xs = np.stack([np.full((self.n, 3), float(self.t + (i + 1) * interval)) for i in range(n_frames)])
        self.t += n_steps
        return xs, np.tile(np.eye(3) * 3.0, (n_frames, 1, 1))


def test_frames_and_leg_layout(tmp_path):
    frames0, boxes0 = run_and_store_frames(FakeContext(7), n_frames=25, steps_per_frame=10, chunk_frames=8)
    assert len(frames0) == 25 and boxes0.shape == (25, 3, 3)
    assert [int(f[0, 0]) for f in frames0] == list(range(10, 260, 10))
    assert frames0._chunk_sizes == [8, 8, 8, 1]
    frames1, boxes1 = run_and_store_frames(FakeContext(7), n_frames=5, steps_per_frame=2)
    fc = FileClient(tmp_path)
    save_leg_results(fc, "solvent", -3.25, 0.4, [0.5, 0.6], 3, (frames0, boxes0), (frames1, boxes1))
    assert sorted(p.name for p in (tmp_path / "solvent").iterdir()) == ["lambda0_traj.npz", "lambda1_traj.npz", "results.npz"]
    res = load_leg_results(fc, "solvent")
    assert float(res["pred_dg"]) == -3.25 and float(res["pred_dg_err"]) == 0.4 and int(res["n_windows"]) == 3
    np.testing.assert_array_equal(res["overlaps"], [0.5, 0.6])
    assert res["lambda0_traj"][0].shape == (25, 7, 3) and res["lambda1_traj"][1].shape == (5, 3, 3)
