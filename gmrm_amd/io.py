"""gmrm's file formats (kept byte for byte): readers for .dim / .bed / .phen / .gri / .grm
and writers for the .bet / .cpn / .csv histories.  File:line references are to the
reference checkout (/root/reference/src)."""
import re
import struct
from pathlib import Path

import numpy as np


def read_dim(path):
    """dimensions.cpp:8-29: one line, two integers N M."""
    with open(path) as f:
        tokens = re.split(r"\s+", f.readline().strip())
    if len(tokens) != 2:
        raise ValueError("dim file should contain a single line with 2 integers")
    return int(tokens[0]), int(tokens[1])


def read_bed(path, N, first=0, n_markers=None, check_magic=True):
    """bayes.cpp:867-900: marker-major 2-bit genotypes after 3 magic bytes (which the
    reference skips unchecked; checked here: 0x6c 0x1b 0x01 = SNP-major)."""
    mbytes = N // 4 if N % 4 == 0 else N // 4 + 1
    with open(path, "rb") as f:
        magic = f.read(3)
        if check_magic and magic != b"\x6c\x1b\x01":
            raise ValueError(f"{path}: not a SNP-major PLINK .bed (magic {magic.hex()})")
        f.seek(3 + first * mbytes)
        count = -1 if n_markers is None else n_markers * mbytes
        raw = np.fromfile(f, dtype=np.uint8, count=count)
    return raw.reshape(-1, mbytes)


def write_bed(path, cols):
    with open(path, "wb") as f:
        f.write(b"\x6c\x1b\x01")
        np.ascontiguousarray(cols, dtype=np.uint8).tofile(f)


def read_phen(path):
    """phenotype.cpp:587-621: whitespace-separated, 3rd token = value or NA -> (y, isna)."""
    ys, nas = [], []
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            tok = re.split(r"\s+", line.rstrip("\n"))
            if tok[2] == "NA":
                ys.append(0.0)
                nas.append(1)
            else:
                ys.append(float(tok[2]))
                nas.append(0)
    return np.array(ys, dtype=np.float64), np.array(nas, dtype=np.uint8)


def read_group_index(path):
    """bayes.cpp:830-853: `label group` per marker, only column 2 is used."""
    out = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if len(tok) >= 2:
                out.append(int(tok[1]))
    return np.array(out, dtype=np.int32)


def read_group_mixture(path):
    """options.cpp:222-286: G lines x K ascending variances, first = 0."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            rows.append([float(x) for x in re.split(r"\s+", line)])
    if len({len(r) for r in rows}) != 1:
        raise ValueError("the same number of mixtures is expected for all groups")
    cva = np.array(rows, dtype=np.float64)
    if np.any(cva[:, 0] != 0.0):
        raise ValueError("first element of group mixture must be 0.0")
    if np.any(np.diff(cva, axis=1) <= 0):
        raise ValueError("mixtures must be given in ascending order")
    return cva


class HistoryWriter:
    """write_ofile_h1<T> (xfiles.hpp:14-38): u32 Mtot | per saved iteration: u32 it + Mtot values."""

    def __init__(self, path, Mtot, dtype):
        self.path, self.Mtot, self.dtype = Path(path), int(Mtot), np.dtype(dtype)
        if self.path.exists():
            self.path.unlink()                      # phenotype.cpp delete_output_files
        with open(self.path, "wb") as f:
            f.write(struct.pack("<I", self.Mtot))

    def write(self, it, n_thinned_saved, values, S=0):
        values = np.ascontiguousarray(values, dtype=self.dtype)
        rec = 4 + self.Mtot * self.dtype.itemsize
        with open(self.path, "r+b") as f:
            if S == 0:
                f.seek(4 + n_thinned_saved * rec)
                f.write(struct.pack("<I", int(it)))
            f.seek(4 + 4 + n_thinned_saved * rec + S * self.dtype.itemsize)
            f.write(values.tobytes())


class CsvWriter:
    """write_ofile_csv (xfiles.cpp:6-47): records of constant length at offset n*len."""

    def __init__(self, path):
        self.path = Path(path)
        if self.path.exists():
            self.path.unlink()
        self.path.touch()

    def write(self, n_thinned_saved, line: bytes):
        with open(self.path, "r+b") as f:
            f.seek(n_thinned_saved * len(line))
            f.write(line)


def read_history(path, dtype):
    """Inverse of HistoryWriter: (Mtot, iterations[], values[n, Mtot])."""
    dtype = np.dtype(dtype)
    raw = Path(path).read_bytes()
    Mtot = struct.unpack_from("<I", raw, 0)[0]
    rec = 4 + Mtot * dtype.itemsize
    n = (len(raw) - 4) // rec
    its = np.array([struct.unpack_from("<I", raw, 4 + i * rec)[0] for i in range(n)], dtype=np.uint32)
    vals = np.array([np.frombuffer(raw, dtype=dtype, count=Mtot, offset=8 + i * rec) for i in range(n)])
    return Mtot, its, vals


def extract_non_zero(path, rec_min, rec_max):
    """What bin/extract_non_zero_betaAll prints (upstream: example/extract_non_zero_betaAll, binary only): for
    every saved record r in [rec_min, rec_max] (0-based position in the .bet file, not the stored iteration
    number) the non-zero effects as (r, marker, beta).  Stops at the last complete record."""
    import struct
    out = []
    with open(path, "rb") as f:
        head = f.read(4)
        if len(head) < 4:
            return out
        (M,) = struct.unpack("<I", head)
        for r in range(max(0, int(rec_min)), int(rec_max) + 1):
            f.seek(4 + r * (4 + 8 * M) + 4)
            raw = f.read(8 * M)
            if len(raw) < 8 * M:
                break
            beta = np.frombuffer(raw, dtype="<f8")
            for m in np.flatnonzero(beta):
                out.append((r, int(m), float(beta[m])))
    return out
