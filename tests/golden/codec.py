"""Compact encodings shared by the golden-vector generator and the tests.

* count maps: the reference stores ``visited`` / ``total`` as float64 arrays of
  small integers; fixtures hold them as one uint32 per cell
  (visited << 16 | total), which deflates to a few hundred KB.
* search fields (``probSP``): almost every cell is exactly 0 or exactly the
  field minimum; fixtures hold a uint8 class image (0 = zero, 1 = minimum,
  2 = other) plus the float64 values of the class-2 cells in row-major order,
  so decoding is bit-exact.
"""
import numpy as np


def pack_counts(visited, total):
    v = np.asarray(visited)
    t = np.asarray(total)
    assert np.array_equal(v, np.rint(v)) and np.array_equal(t, np.rint(t))
    assert v.max() < 65536 and t.max() < 65536 and v.min() >= 0 and t.min() >= 0
    return (v.astype(np.uint32) << np.uint32(16)) | t.astype(np.uint32)


def unpack_counts(packed):
    p = np.asarray(packed, dtype=np.uint32)
    return (p >> np.uint32(16)).astype(np.float64), (p & np.uint32(0xFFFF)).astype(np.float64)


def encode_field(field):
    f = np.asarray(field, dtype=np.float64)
    floor = f.min()
    cls = np.full(f.shape, 2, dtype=np.uint8)
    cls[f == 0] = 0
    cls[f == floor] = 1
    return dict(cls=cls, floor=np.float64(floor), other=f[cls == 2].copy())


def decode_field(cls, floor, other):
    f = np.zeros(cls.shape, dtype=np.float64)
    f[cls == 1] = floor
    f[cls == 2] = other
    return f


def nan_if_none(v):
    return np.float64(np.nan) if v is None else np.float64(v)


def none_if_nan(v):
    v = float(v)
    return None if np.isnan(v) else v
