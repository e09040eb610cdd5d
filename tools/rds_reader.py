"""Minimal reader for R's XDR serialisation (format version 3, as written by saveRDS).

Only what the stored MAGMA training files need: NULL, symbols, pairlists (attributes),
character / integer / logical / real / string vectors and generic lists. Decoding stops with
``StopDecoding`` at the first closure / byte-code object (the stored ``kern_0`` / ``kern_i``),
which come after every dataset entry in ``Simulations/Training/train_*.rds``
(written at simulation_study.R:257-258, :543).

Nothing from the file is executed: this is a pure byte parser (no pickle, no eval).
"""
import gzip
import struct


class StopDecoding(Exception):
    pass


class RObj:
    """A decoded R vector/list with its attributes."""

    def __init__(self, value, attrs=None):
        self.value = value
        self.attrs = attrs or {}

    def names(self):
        n = self.attrs.get("names")
        return None if n is None else n.value

    def as_dict(self):
        return dict(zip(self.names(), self.value))

    def __repr__(self):
        return f"RObj({self.value!r}, attrs={list(self.attrs)})"


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.p = 0
        self.refs = []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def f64(self, n):
        v = struct.unpack_from(f">{n}d", self.b, self.p)
        self.p += 8 * n
        return list(v)

    def ints(self, n):
        v = struct.unpack_from(f">{n}i", self.b, self.p)
        self.p += 4 * n
        return list(v)

    def item(self):
        flags = self.i32()
        typ = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if typ == 254:  # NILVALUE
            return None
        if typ == 255:  # REFSXP
            return self.refs[(flags >> 8) - 1]
        if typ == 1:  # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if typ == 9:  # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            s = self.b[self.p:self.p + n].decode("utf-8")
            self.p += n
            return s
        if typ == 2:  # LISTSXP (pairlist) -> list of (tag, value)
            out = []
            while True:
                attrs = self.item() if has_attr else None  # noqa: F841
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                nflags = self.i32()
                ntyp = nflags & 0xFF
                if ntyp == 254:
                    break
                if ntyp != 2:
                    raise ValueError(f"unexpected pairlist tail type {ntyp}")
                has_attr = bool(nflags & (1 << 9))
                has_tag = bool(nflags & (1 << 10))
            return out
        if typ in (10, 13):  # LGLSXP / INTSXP
            n = self.i32()
            val = self.ints(n)
        elif typ == 14:  # REALSXP
            n = self.i32()
            val = self.f64(n)
        elif typ == 16:  # STRSXP
            n = self.i32()
            val = [self.item() for _ in range(n)]
        elif typ == 19:  # VECSXP
            n = self.i32()
            val = []
            for _ in range(n):
                val.append(self.item())
        else:
            raise StopDecoding(f"type {typ} at byte {self.p}")
        attrs = {}
        if has_attr:
            for tag, v in self.item():
                attrs[tag] = v
        return RObj(val, attrs)


def read_rds_toplevel_list(path, max_items=None):
    """Return (names, items) of the top-level named list, reading items until the first
    undecodable element (closure) or ``max_items``."""
    buf = gzip.open(path).read()
    assert buf[:2] == b"X\n", "not an XDR-serialised RDS"
    r = _Reader(buf)
    r.p = 2
    version = r.i32()
    writer = r.i32()
    r.i32()  # min reader version
    if version == 3:
        n = r.i32()
        r.p += n  # native encoding string
    flags = r.i32()
    assert flags & 0xFF == 19, "top level is not a list"
    has_attr = bool(flags & (1 << 9))
    n = r.i32()
    items = []
    for k in range(n):
        if max_items is not None and k >= max_items:
            break
        try:
            items.append(r.item())
        except StopDecoding:
            break
    return {"version": version, "writer": hex(writer), "n": n, "has_attr": has_attr}, items
