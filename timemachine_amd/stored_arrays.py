"""Frame storage and the on-disk result layout that sits downstream of `Context.multiple_steps`.

Mirrors: `StoredArrays` (reference `timemachine/fe/stored_arrays.py:14-133`: a sequence of numpy
arrays with O(1) memory, one `.npy` file per appended chunk in a private temporary directory, `store` / `load` through a
file client), `serialize_array` / `deserialize_array` (:136-146), `FileClient` (`timemachine/parallel/client.py:341-370`)
and the per-leg files `examples/run_rbfe_legs.py:140-160` writes (`results.npz`: pred_dg, pred_dg_err, overlaps,
n_windows; `lambda0_traj.npz` / `lambda1_traj.npz`: coords, boxes) -- what downstream analysis reads.
Host-side Python, as in the reference."""

import io
import tempfile
from collections.abc import Sequence
from itertools import count
from pathlib import Path

import numpy as np


class FileClient:
    """files under a base directory, addressed by relative path"""

    def __init__(self, base=None):
        self.base = Path(base) if base is not None else Path.cwd()

    def full_path(self, path):
        return str(Path(self.base, path).absolute())

    def exists(self, path):
        return Path(self.full_path(path)).exists()

    def store(self, path, data: bytes):
        p = Path(self.full_path(path))
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)

    def store_stream(self, path, stream):
        p = Path(self.full_path(path))
        p.parent.mkdir(parents=True, exist_ok=True)
        with open(p, "wb") as out:
            while True:
                block = stream.read(io.DEFAULT_BUFFER_SIZE)
                if not block:
                    break
                out.write(block)

    def load(self, path) -> bytes:
        return Path(self.full_path(path)).read_bytes()

    def delete(self, path):
        Path(self.full_path(path)).unlink()


def serialize_array(array) -> bytes:
    buf = io.BytesIO()
    np.save(buf, array)
    return buf.getvalue()


def deserialize_array(bs: bytes):
    return np.load(io.BytesIO(bs))


class StoredArrays(Sequence):
    """Sequence of arrays kept on disk: `extend(xs)` writes one chunk file; iteration and integer indexing read chunk
    files back one at a time.  The temporary directory disappears with the object."""

    def __init__(self):
        self._chunk_sizes = []
        self._dir = tempfile.TemporaryDirectory()

    @classmethod
    def from_chunks(cls, chunks):
        sa = cls()
        for chunk in chunks:
            sa.extend(chunk)
        return sa

    @staticmethod
    def get_chunk_path(path, idx):
        return (Path(path) / str(idx)).with_suffix(".npy")

    def _chunk_file(self, idx):
        return self.get_chunk_path(self._dir.name, idx)

    def _chunks(self):
        for idx in range(len(self._chunk_sizes)):
            yield np.load(self._chunk_file(idx))

    def extend(self, xs):
        np.save(self._chunk_file(len(self._chunk_sizes)), np.asarray(xs))
        self._chunk_sizes.append(len(xs))

    def __len__(self):
        return sum(self._chunk_sizes)

    def __iter__(self):
        for chunk in self._chunks():
            yield from chunk

    def __getitem__(self, key):
        if isinstance(key, slice):
            raise NotImplementedError("slices are not implemented")
        if not isinstance(key, (int, np.integer)):
            raise ValueError("invalid subscript")
        pos = range(len(self))[key]  # normalises negatives, raises IndexError out of range
        for idx, size in enumerate(self._chunk_sizes):
            if pos < size:
                return np.load(self._chunk_file(idx))[pos]
            pos -= size
        raise AssertionError("internal error")

    def __eq__(self, other):
        return self._chunk_sizes == other._chunk_sizes and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(self, other))

    def __reduce__(self):
        return self.from_chunks, (list(self._chunks()),)

    def store(self, client, prefix=Path(".")):
        """copy the chunk files to persistent storage (refuses to overwrite)"""
        for idx in range(len(self._chunk_sizes)):
            dest = self.get_chunk_path(prefix, idx)
            if client.exists(str(dest)):
                raise FileExistsError(f"file already exists: {dest}")
            with open(self._chunk_file(idx), "rb") as src:
                client.store_stream(str(dest), src)

    @classmethod
    def load(cls, client, prefix=Path(".")):
        sa = cls()
        for idx in count():
            path = cls.get_chunk_path(prefix, idx)
            if not client.exists(str(path)):
                break
            sa.extend(list(deserialize_array(client.load(str(path)))))
        return sa


def run_and_store_frames(ctxt, n_frames, steps_per_frame, chunk_frames=100):
    """Drive `Context.multiple_steps` in chunks and keep the frames off the heap: returns (StoredArrays of [N,3] frames,
    boxes [n_frames,3,3]).  One D2H copy per stored frame, as in the reference's sampling loops."""
    frames = StoredArrays()
    boxes = []
    done = 0
    while done < n_frames:
        take = min(chunk_frames, n_frames - done)
        xs, bs = ctxt.multiple_steps(take * steps_per_frame, steps_per_frame)
        frames.extend(xs)
        boxes.extend(bs)
        done += take
    return frames, np.asarray(boxes)


def save_leg_results(client, leg_name, pred_dg, pred_dg_err, overlaps, n_windows, traj0, traj1):
    """write results.npz / lambda0_traj.npz / lambda1_traj.npz for one leg; traj = (frames, boxes)"""
    leg = Path(leg_name)
    Path(client.full_path(leg)).mkdir(parents=True, exist_ok=True)
    np.savez_compressed(
        client.full_path(leg / "results.npz"),
        pred_dg=float(pred_dg),
        pred_dg_err=float(pred_dg_err),
        overlaps=np.asarray(overlaps),
        n_windows=int(n_windows),
    )
    for name, (frames, boxes) in (("lambda0_traj.npz", traj0), ("lambda1_traj.npz", traj1)):
        np.savez_compressed(client.full_path(leg / name), coords=np.array(list(frames)), boxes=np.asarray(boxes))


def load_leg_results(client, leg_name):
    leg = Path(leg_name)
    out = {k: v for k, v in np.load(client.full_path(leg / "results.npz")).items()}
    for name in ("lambda0_traj", "lambda1_traj"):
        t = np.load(client.full_path(leg / f"{name}.npz"))
        out[name] = (t["coords"], t["boxes"])
    return out
