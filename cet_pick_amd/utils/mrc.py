"""Dependency-free MRC2014 reader / writer.

Mirror of the reference's `cet_pick/utils/mrc.py` API (`MRCHeader`, `parse_header`, `parse_mrc`, `write`,
:20-172) plus `open_data`, the replacement for `mrcfile.open(path, permissive=True).data` used by
`utils/loader.py:29-30` (the `mrcfile` package is an un-vendored third-party dependency).  The 1024-byte
header layout is the MRC2014 / IMOD layout the reference's struct string describes; it is expressed here
as a numpy structured dtype.
"""
import os

import numpy as np

# MRC mode -> numpy element type (mode 3 / 4 complex and 16 RGB are listed for completeness)
DTYPE_FOR_MODE = {0: np.dtype(np.int8), 1: np.dtype(np.int16), 2: np.dtype(np.float32),
                  3: np.dtype([("re", np.int16), ("im", np.int16)]), 4: np.dtype(np.complex64),
                  6: np.dtype(np.uint16), 12: np.dtype(np.float16),
                  16: np.dtype([("r", np.uint8), ("g", np.uint8), ("b", np.uint8)])}
MODE_FOR_DTYPE = {np.dtype(np.int8): 0, np.dtype(np.int16): 1, np.dtype(np.float32): 2,
                  np.dtype(np.complex64): 4, np.dtype(np.uint16): 6, np.dtype(np.float16): 12}

_HEADER_LAYOUT = [
    ("nx", "i4"), ("ny", "i4"), ("nz", "i4"), ("mode", "i4"),
    ("nxstart", "i4"), ("nystart", "i4"), ("nzstart", "i4"),
    ("mx", "i4"), ("my", "i4"), ("mz", "i4"),
    ("xlen", "f4"), ("ylen", "f4"), ("zlen", "f4"),
    ("alpha", "f4"), ("beta", "f4"), ("gamma", "f4"),
    ("mapc", "i4"), ("mapr", "i4"), ("maps", "i4"),
    ("amin", "f4"), ("amax", "f4"), ("amean", "f4"),
    ("ispg", "i4"), ("next", "i4"), ("creatid", "i2"), ("_pad0", "V30"),
    ("nint", "i2"), ("nreal", "i2"), ("_pad1", "V20"),
    ("imodStamp", "i4"), ("imodFlags", "i4"),
    ("idtype", "i2"), ("lens", "i2"), ("nd1", "i2"), ("nd2", "i2"), ("vd1", "i2"), ("vd2", "i2"),
    ("tilt_ox", "f4"), ("tilt_oy", "f4"), ("tilt_oz", "f4"),
    ("tilt_cx", "f4"), ("tilt_cy", "f4"), ("tilt_cz", "f4"),
    ("xorg", "f4"), ("yorg", "f4"), ("zorg", "f4"),
    ("cmap", "S4"), ("stamp", "V4"), ("rms", "f4"),
    ("nlabl", "i4"), ("labels", "V800"),
]
HEADER_DTYPE_LE = np.dtype(_HEADER_LAYOUT).newbyteorder("<")
HEADER_DTYPE_BE = np.dtype(_HEADER_LAYOUT).newbyteorder(">")
assert HEADER_DTYPE_LE.itemsize == 1024


class MRCHeader:
    """The fixed header as a dict-like `fields` plus the raw extended header bytes."""

    FIELDS = [n for n, _ in _HEADER_LAYOUT if not n.startswith("_pad")]

    def __init__(self, record, extended_header=b"", big_endian=False):
        self.record = record.copy()
        self.extended_header = extended_header
        self.big_endian = big_endian

    @property
    def fields(self):
        return {k: self.record[k].item() if self.record[k].dtype.kind in "if" else bytes(self.record[k])
                for k in self.FIELDS}

    def __getitem__(self, k):
        v = self.record[k]
        return v.item() if v.dtype.kind in "if" else bytes(v)

    def __setitem__(self, k, v):
        self.record[k] = v

    @property
    def D(self):
        return self["nx"]

    def __str__(self):
        return "Header: %s\nExtended header: %d bytes" % (self.fields, len(self.extended_header))

    @classmethod
    def parse(cls, fname):
        with open(fname, "rb") as f:
            raw = f.read(1024)
            if len(raw) < 1024:
                raise ValueError("%s: shorter than an MRC header" % fname)
            rec = np.frombuffer(raw, dtype=HEADER_DTYPE_LE, count=1)[0]
            big = False
            # machine stamp 0x11 0x11 = big endian; otherwise judge by a sane mode / nx
            if not (0 <= rec["mode"] <= 1024 and 0 < rec["nx"] < (1 << 24)):
                rec = np.frombuffer(raw, dtype=HEADER_DTYPE_BE, count=1)[0]
                big = True
            ext = f.read(max(int(rec["next"]), 0))
        return cls(rec, ext, big)

    @classmethod
    def make_default_header(cls, data, is_vol=True, Apix=1., xorg=0., yorg=0., zorg=0.):
        nz, ny, nx = data.shape
        rec = np.zeros((), dtype=HEADER_DTYPE_LE)
        rec["nx"], rec["ny"], rec["nz"] = nx, ny, nz
        rec["mode"] = MODE_FOR_DTYPE.get(np.dtype(data.dtype), 2)
        rec["mx"], rec["my"], rec["mz"] = nx, ny, nz
        rec["xlen"], rec["ylen"], rec["zlen"] = Apix * nx, Apix * ny, Apix * nz
        rec["alpha"] = rec["beta"] = rec["gamma"] = 90.
        rec["mapc"], rec["mapr"], rec["maps"] = 1, 2, 3
        if is_vol:       # volumes carry real statistics, image stacks the "undefined" convention
            rec["amin"], rec["amax"], rec["amean"], rec["rms"] = data.min(), data.max(), data.mean(), data.std()
            rec["ispg"] = 1
            rec["cmap"] = b"MAP "
        else:
            rec["amin"], rec["amax"], rec["amean"], rec["rms"] = -1, -2, -3, -1
        rec["xorg"], rec["yorg"], rec["zorg"] = xorg, yorg, zorg
        return cls(rec)

    def write(self, fh):
        fh.write(self.record.tobytes())
        fh.write(self.extended_header)

    def get_apix(self):
        return self["xlen"] / self["nx"]

    def update_apix(self, Apix):
        for ax in "xyz":
            self[ax + "len"] = self["n" + ax] * Apix

    def get_origin(self):
        return self["xorg"], self["yorg"], self["zorg"]

    def update_origin(self, xorg, yorg, zorg):
        self["xorg"], self["yorg"], self["zorg"] = xorg, yorg, zorg


class LazyImage:
    """One (ny, nx) section, read on demand."""

    def __init__(self, fname, shape, dtype, offset):
        self.fname, self.shape, self.dtype, self.offset = fname, shape, dtype, offset

    def get(self):
        return np.fromfile(self.fname, dtype=self.dtype, count=int(np.prod(self.shape)),
                           offset=self.offset).reshape(self.shape)


def parse_header(fname):
    return MRCHeader.parse(fname)


def _data_dtype(header):
    mode = header["mode"]
    if mode not in DTYPE_FOR_MODE:
        raise ValueError("unsupported MRC mode %d" % mode)
    return DTYPE_FOR_MODE[mode].newbyteorder(">" if header.big_endian else "<")


def parse_mrc(fname, lazy=False):
    """(array (nz, ny, nx) | list of LazyImage, header)."""
    header = MRCHeader.parse(fname)
    start = 1024 + max(header["next"], 0)
    dtype = _data_dtype(header)
    nz, ny, nx = header["nz"], header["ny"], header["nx"]
    if lazy:
        stride = dtype.itemsize * ny * nx
        return [LazyImage(fname, (ny, nx), dtype, start + i * stride) for i in range(nz)], header
    avail = (os.path.getsize(fname) - start) // (dtype.itemsize * ny * nx)
    if avail < nz:       # permissive: a truncated file yields the complete sections it holds
        nz = int(avail)
    arr = np.fromfile(fname, dtype=dtype, count=nz * ny * nx, offset=start).reshape((nz, ny, nx))
    return arr, header


def open_data(path, mmap=False):
    """`mrcfile.open(path, permissive=True).data`: the (nz, ny, nx) array in the file's element type."""
    if mmap:
        header = MRCHeader.parse(path)
        start = 1024 + max(header["next"], 0)
        return np.memmap(path, dtype=_data_dtype(header), mode="r", offset=start,
                         shape=(header["nz"], header["ny"], header["nx"]))
    return parse_mrc(path)[0]


def parse_mrc_list(txtfile, lazy=False):
    base = os.path.dirname(os.path.abspath(txtfile))
    names = [ln.strip() for ln in open(txtfile) if ln.strip()]
    names = [n if os.path.isabs(n) else os.path.join(base, n) for n in names]
    if lazy:
        return [img for n in names for img in parse_mrc(n, lazy=True)[0]]
    return np.vstack([parse_mrc(n)[0] for n in names])


def write(fname, array, header=None, Apix=1., xorg=0., yorg=0., zorg=0., is_vol=None):
    array = np.ascontiguousarray(array)
    if header is None:
        if is_vol is None:
            is_vol = len(set(array.shape)) == 1      # the reference's guess: a cube is a volume
        header = MRCHeader.make_default_header(array, is_vol, Apix, xorg, yorg, zorg)
    with open(fname, "wb") as f:
        header.write(f)
        f.write(memoryview(array).cast("B"))     # the payload straight from the (contiguous) array: no tobytes() copy
