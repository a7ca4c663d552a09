"""clpio: the named-flat-array container spoken by oracle/ref/harness.c.

file   = b"CLPIO1\\0\\0", u64 count, records
record = char name[24], u64 nbytes, payload padded to 8 bytes
"""
import struct

import numpy as np

MAGIC = b"CLPIO1\0\0"


def write(path, arrays):
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            raw = a.tobytes()
            nm = name.encode()
            assert len(nm) < 24, name
            f.write(nm.ljust(24, b"\0"))
            f.write(struct.pack("<Q", len(raw)))
            f.write(raw)
            f.write(b"\0" * ((-len(raw)) % 8))


def read(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(8) == MAGIC, "bad clpio magic"
        (count,) = struct.unpack("<Q", f.read(8))
        for _ in range(count):
            name = f.read(24).split(b"\0", 1)[0].decode()
            (nbytes,) = struct.unpack("<Q", f.read(8))
            out[name] = f.read(nbytes)
            f.read((-nbytes) % 8)
    return out


def as_array(raw, dtype, shape=None):
    a = np.frombuffer(raw, dtype=dtype).copy()
    return a.reshape(shape) if shape is not None else a
