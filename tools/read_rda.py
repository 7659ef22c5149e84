"""Minimal reader of R's .rda / .rds serialisation (XDR, format version 2 or 3) — enough for numeric / integer /
logical / character vectors, lists, pairlists, factors, data.frames and S4 objects whose slots are those (terra's
PackedSpatRaster).  Written from R's documented serialisation format (R Internals, "Serialization Formats"); used to
turn the reference package's bundled example DATA into .npz test fixtures (tests/golden/make_bundled_inputs.py).

    objs = read_rda(path)      # {name: value}; vectors -> numpy arrays, lists -> RList (dict-like with .attrs)
"""
import bz2
import gzip
import lzma
import struct

import numpy as np


class RObj:
    """value + attributes"""
    def __init__(self, value, attrs=None, kind=""):
        self.value, self.attrs, self.kind = value, attrs or {}, kind

    def __repr__(self):
        v = self.value
        d = f"{type(v).__name__}" + (f"{getattr(v, 'shape', '')}" if hasattr(v, "shape") else f"[{len(v)}]" if hasattr(v, "__len__") else "")
        return f"RObj<{self.kind} {d} attrs={list(self.attrs)}>"

    def names(self):
        n = self.attrs.get("names")
        return [] if n is None else list(n.value)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.value[self.names().index(key)]
        return self.value[key]


class _Reader:
    def __init__(self, b):
        self.b, self.p, self.refs = b, 0, []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def raw(self, n):
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) + (lo & 0xffffffff)
        return n

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_obj, has_attr, has_tag = bool(flags & 0x100), bool(flags & 0x200), bool(flags & 0x400)
        if t == 254:                      # NILVALUE
            return None
        if t in (253, 252, 251, 250, 242, 241):   # global/empty/base env, missing arg, base/unbound
            return RObj(None, kind=f"special{t}")
        if t == 255:                      # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:                        # SYMSXP
            name = self.item()
            self.refs.append(name)
            return name
        if t == 9:                        # CHARSXP
            n = self.i32()
            return None if n == -1 else self.raw(n).decode("utf-8", "replace")
        if t in (2, 6):                   # LISTSXP / LANGSXP: walk the chain iteratively
            out, attrs = [], None
            while True:
                a = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                if a is not None and attrs is None:
                    attrs = a
                flags = self.i32()
                t2 = flags & 0xFF
                if t2 == 254:
                    break
                if t2 not in (2, 6):
                    raise ValueError(f"unexpected cdr type {t2}")
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
            return RObj(out, kind="pairlist")
        if t == 4:                        # ENVSXP: locked, enclos, frame, hashtab, attrib
            env = RObj({}, kind="env")
            self.refs.append(env)
            self.i32()
            for _ in range(4):
                self.item()
            return env
        if t == 10 or t == 13:            # LGLSXP / INTSXP
            n = self.length()
            v = np.frombuffer(self.raw(4 * n), dtype=">i4").astype(np.int32)
            obj = RObj(v, kind="logical" if t == 10 else "integer")
        elif t == 14:                     # REALSXP
            n = self.length()
            obj = RObj(np.frombuffer(self.raw(8 * n), dtype=">f8").astype(np.float64), kind="double")
        elif t == 16:                     # STRSXP
            n = self.length()
            obj = RObj([self.item() for _ in range(n)], kind="character")
        elif t == 19 or t == 20:          # VECSXP / EXPRSXP
            n = self.length()
            obj = RObj([self.item() for _ in range(n)], kind="list")
        elif t == 24:                     # RAWSXP
            n = self.length()
            obj = RObj(self.raw(n), kind="raw")
        elif t == 25:                     # S4SXP
            obj = RObj(None, kind="S4")
        elif t == 238:                    # ALTREP: info, state, attr -> only the compact sequences / wrappers
            info, state, attr = self.item(), self.item(), self.item()
            cls = info.value[0][1]
            if cls in ("compact_intseq", "compact_realseq"):
                n, start, step = (float(x) for x in state.value)
                v = start + step * np.arange(int(n))
                obj = RObj(v.astype(np.int32) if cls == "compact_intseq" else v, kind="altrep")
            elif cls.startswith("wrap_"):
                obj = state.value[0] if isinstance(state.value, list) else state
            else:
                raise ValueError(f"ALTREP class {cls} not handled")
            if attr is not None:
                obj.attrs.update({k: v for k, v in attr.value})
            return obj
        else:
            raise ValueError(f"SEXPTYPE {t} not handled at byte {self.p}")
        if has_attr:
            a = self.item()
            if a is not None:
                obj.attrs = {k: v for k, v in a.value}
        return obj


def _decompress(b):
    if b[:2] == b"\x1f\x8b":
        return gzip.decompress(b)
    if b[:3] == b"BZh":
        return bz2.decompress(b)
    if b[:6] == b"\xfd7zXZ\x00":
        return lzma.decompress(b)
    return b


def _body(b):
    r = _Reader(b)
    if r.raw(2) != b"X\n":
        raise ValueError("only XDR serialisation is handled")
    version = r.i32()
    r.i32(); r.i32()
    if version == 3:
        r.raw(r.i32())
    return r.item()


def read_rda(path):
    b = _decompress(open(path, "rb").read())
    if b[:5] not in (b"RDX2\n", b"RDX3\n"):
        raise ValueError("not an .rda file")
    top = _body(b[5:])
    return {k: v for k, v in top.value}


def read_rds(path):
    return _body(_decompress(open(path, "rb").read()))


if __name__ == "__main__":
    import sys
    for name, obj in read_rda(sys.argv[1]).items():
        print(name, obj)
        if isinstance(obj, RObj) and obj.kind == "list":
            for n, v in zip(obj.names(), obj.value):
                print("   ", n, v)
        if isinstance(obj, RObj) and obj.kind == "S4":
            for k, v in obj.attrs.items():
                print("    @", k, v)
