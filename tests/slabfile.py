"""Reader/writer for the SLAB0001 named-array container (oracle/slabio.h).

Golden fixtures under tests/golden/ are stored in this format (optionally
gzip-compressed).  Pure numpy; nothing is executed from the file.
"""
import gzip
import hashlib
import struct

import numpy as np

_DT = {0: np.int32, 1: np.int64, 2: np.uint64, 3: np.float64}
_CODE = {np.dtype(v): k for k, v in _DT.items()}


def load(path):
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        buf = f.read()
    if buf[:8] != b"SLAB0001":
        raise ValueError(f"{path}: not a slab file")
    out, off = {}, 8
    while off < len(buf):
        name = buf[off:off + 24].split(b"\0", 1)[0].decode()
        dtype, _ = struct.unpack_from("<ii", buf, off + 24)
        (count,) = struct.unpack_from("<q", buf, off + 32)
        off += 40
        dt = np.dtype(_DT[dtype])
        nbytes = dt.itemsize * count
        out[name] = np.frombuffer(buf, dtype=dt, count=count, offset=off).copy()
        off += nbytes + (8 - nbytes % 8) % 8
    return out


def save(path, arrays):
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(b"SLAB0001")
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            code = _CODE[a.dtype]
            f.write(name.encode().ljust(24, b"\0")[:24])
            f.write(struct.pack("<iiq", code, 0, a.size))
            raw = a.tobytes()
            f.write(raw)
            f.write(b"\0" * ((8 - len(raw) % 8) % 8))


FACTOR_KEYS = ("pinv", "Lp", "Li", "Llen", "Llimbs", "Up", "Ui", "Ulen", "Ulimbs",
               "rholen", "rholimbs")


def factor_digest(d):
    """SHA-256 over the canonical factor arrays (original row ids)."""
    h = hashlib.sha256()
    for k in FACTOR_KEYS:
        a = np.ascontiguousarray(d[k])
        if k in ("Lp", "Up"):
            a = a.astype(np.int64)
        h.update(k.encode())
        h.update(a.tobytes())
    return h.hexdigest()
