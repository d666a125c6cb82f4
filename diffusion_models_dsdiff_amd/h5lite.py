"""HDF5 slice files without h5py — numpy + zlib only.

The reference keeps every 2-D slice in its own HDF5 file: ``preprocess/to_h5.py:40-50`` writes ``f[key] = array`` for the keys
``F_Data1, F_Data2, S_Data1, S_Data2`` (BraTS: one key per modality), and ``LoadH5`` (``training_project/utils/
my_transform.py:142-154``) reads them back with ``h5py.File(path)[key][()]``.  h5py is not installed for the interpreter
this package runs under, so the container is read (and written) here from its public specification ("HDF5 File Format
Specification Version 3.0"):

  read_h5(path) / H5File(path)   superblock v0-v3 (with or without a user block); groups as symbol tables (B-tree v1 + local
                                 heap + SNOD: what ``h5py.File(..., 'w')`` writes by default) and as compact link messages
                                 (``libver='latest'``, up to 8 links); object headers v1 and v2 with continuation blocks;
                                 datasets contiguous, compact or chunked through a v1 B-tree (``compression='gzip'``,
                                 ``shuffle``, ``fletcher32``), layout message v1-v3 and v4 (contiguous / compact / single chunk / implicit /
                                 fixed-array chunk index);
                                 fixed-point and IEEE floating-point types of 1-8 bytes, either byte order.
                                 Anything else (dense link storage, extensible-array / v2 B-tree chunk indices, compound / string / variable-
                                 length types, external or virtual storage) raises NotImplementedError naming the feature.
  write_h5(path, arrays)         the file ``to_h5.py`` produces: superblock v0, one root symbol table, contiguous datasets.
  LoadH5(path_key, keys)         the reference's dictionary transform, same arguments and effect.

Pinned by files that real h5py 3.3 / libhdf5 1.10.6 wrote the way the reference does (tests/golden/h5/, generator
tests/golden/gen_h5.py) and, the other way round, by h5py reading what ``write_h5`` wrote (tests/test_h5lite.py).
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(ValueError):
    pass


def _unsupported(what: str):
    raise NotImplementedError(f"h5lite: {what} is outside the subset this reader covers (see the module docstring)")


# ===================================================================================================== reader
class _Dataset:
    def __init__(self, f: "H5File", name: str, msgs: List[Tuple[int, int, bytes]]):
        self.file, self.name = f, name
        self.shape: Tuple[int, ...] = ()
        self.dtype: Optional[np.dtype] = None
        self.layout = None
        self.filters: List[Tuple[int, Tuple[int, ...]]] = []
        for mtype, _flags, body in msgs:
            if mtype == 0x01:
                self.shape = self._dataspace(body)
            elif mtype == 0x03:
                self.dtype = self._datatype(body)
            elif mtype == 0x08:
                self.layout = body
            elif mtype == 0x0B:
                self.filters = self._filters(body)
        if self.dtype is None or self.layout is None:
            raise H5FormatError(f"'{name}' is not a dataset (no datatype / layout message)")

    # ---- header messages
    def _dataspace(self, b: bytes) -> Tuple[int, ...]:
        ver, rank, flags = b[0], b[1], b[2]
        if ver == 1:
            off = 8
        elif ver == 2:
            if b[3] == 2:   # null dataspace
                return (0,)
            off = 4
        else:
            raise H5FormatError(f"dataspace message version {ver}")
        L = self.file.L
        return tuple(int.from_bytes(b[off + i * L: off + (i + 1) * L], "little") for i in range(rank))

    @staticmethod
    def _datatype(b: bytes) -> np.dtype:
        cls, ver = b[0] & 0x0F, b[0] >> 4
        bits0 = b[1]
        size = struct.unpack_from("<I", b, 4)[0]
        order = ">" if bits0 & 1 else "<"
        if cls == 0:     # fixed point
            if size not in (1, 2, 4, 8):
                _unsupported(f"{size}-byte integer type")
            return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}")
        if cls == 1:     # floating point: the IEEE layouts only
            if size not in (2, 4, 8):
                _unsupported(f"{size}-byte floating-point type")
            prec, eloc, esize, mloc, msize = struct.unpack_from("<HBBBB", b, 10)
            want = {2: (16, 10, 5, 0, 10), 4: (32, 23, 8, 0, 23), 8: (64, 52, 11, 0, 52)}[size]
            if (prec, eloc, esize, mloc, msize) != want:
                _unsupported("non-IEEE floating-point layout")
            return np.dtype(f"{order}f{size}")
        names = {2: "time", 3: "string", 4: "bit field", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 9: "variable-length",
                 10: "array"}
        _unsupported(f"datatype class {cls} ({names.get(cls, '?')}, version {ver})")

    @staticmethod
    def _filters(b: bytes) -> List[Tuple[int, Tuple[int, ...]]]:
        ver, n = b[0], b[1]
        off = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", b, off)[0]
            off += 2
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", b, off)[0]
                off += 2
            else:
                nlen = 0
            _fl, nvals = struct.unpack_from("<HH", b, off)
            off += 4
            off += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            vals = struct.unpack_from(f"<{nvals}I", b, off)
            off += 4 * nvals
            if ver == 1 and nvals % 2:
                off += 4
            out.append((fid, vals))
        return out

    # ---- data
    def read(self) -> np.ndarray:
        f = self.file
        b = self.layout
        ver = b[0]
        count = int(np.prod(self.shape)) if self.shape else 1
        nbytes = count * self.dtype.itemsize
        if ver in (1, 2):
            rank, cls = b[1], b[2]
            off = 8
            addr = None
            if cls != 0:
                addr = f._addr(b, off)
                off += f.O
            dims = struct.unpack_from(f"<{rank}I", b, off)
            off += 4 * rank
            if cls == 0:
                size = struct.unpack_from("<I", b, off)[0]
                raw = b[off + 4: off + 4 + size]
            elif cls == 1:
                raw = f._read(addr, nbytes) if addr != UNDEF else bytes(nbytes)
            else:
                return self._chunked(addr, dims)
        elif ver in (3, 4):
            cls = b[1]
            if cls == 0:
                size = struct.unpack_from("<H", b, 2)[0]
                raw = b[4: 4 + size]
            elif cls == 1:
                addr = f._addr(b, 2)
                raw = f._read(addr, nbytes) if addr != UNDEF else bytes(nbytes)
            elif cls == 2 and ver == 3:
                rank = b[2]
                addr = f._addr(b, 3)
                dims = struct.unpack_from(f"<{rank}I", b, 3 + f.O)
                return self._chunked(addr, dims)
            elif cls == 2:
                return self._chunked_v4(b)
            else:
                _unsupported(f"data layout class {cls} (virtual / external storage)")
        else:
            raise H5FormatError(f"data layout message version {ver}")
        return np.frombuffer(raw[:nbytes], dtype=self.dtype).reshape(self.shape).astype(self.dtype.newbyteorder("="))

    def _unfilter(self, raw: bytes, mask: int) -> bytes:
        for i in reversed(range(len(self.filters))):
            if mask >> i & 1:
                continue
            fid, vals = self.filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                es = vals[0] if vals else self.dtype.itemsize
                n = len(raw) // es
                a = np.frombuffer(raw[: n * es], dtype=np.uint8).reshape(es, n)
                raw = a.T.tobytes() + raw[n * es:]
            elif fid == 3:
                raw = raw[:-4]
            else:
                _unsupported(f"filter id {fid}")
        return raw

    def _place(self, out: np.ndarray, raw: bytes, offs: Tuple[int, ...], cdims: Tuple[int, ...]):
        chunk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(cdims))).reshape(cdims)
        sl_o = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
        sl_c = tuple(slice(0, s.stop - s.start) for s in sl_o)
        out[sl_o] = chunk[sl_c]

    def _chunked(self, btree: int, dims: Tuple[int, ...]) -> np.ndarray:
        cdims = tuple(dims[:-1])   # the last entry is the element size
        if len(cdims) != len(self.shape):
            raise H5FormatError("chunk rank does not match the dataspace")
        out = np.zeros(self.shape, dtype=self.dtype.newbyteorder("="))
        if btree == UNDEF:
            return out
        for size, mask, offs, addr in self.file._chunk_btree(btree, len(dims)):
            self._place(out, self._unfilter(self.file._read(addr, size), mask), offs[:-1], cdims)
        return out

    def _chunked_v4(self, b: bytes) -> np.ndarray:
        f = self.file
        flags, rank, enc = b[2], b[3], b[4]
        dims = tuple(int.from_bytes(b[5 + i * enc: 5 + (i + 1) * enc], "little") for i in range(rank))
        off = 5 + rank * enc
        index = b[off]
        off += 1
        cdims = dims[:-1]
        out = np.zeros(self.shape, dtype=self.dtype.newbyteorder("="))
        if index == 1:   # single chunk
            if flags & 2:
                size = int.from_bytes(b[off: off + f.L], "little")
                mask = struct.unpack_from("<I", b, off + f.L)[0]
                off += f.L + 4
            else:
                size, mask = int(np.prod(cdims)) * self.dtype.itemsize, 0xFFFFFFFF
            addr = f._addr(b, off)
            if addr != UNDEF:
                raw = f._read(addr, size)
                raw = self._unfilter(raw, mask) if flags & 2 else raw
                self._place(out, raw, (0,) * len(cdims), cdims)
            return out
        if index == 2:   # implicit: all chunks allocated back to back, no filters
            addr = f._addr(b, off)
            grid = [-(-s // c) for s, c in zip(self.shape, cdims)]
            csize = int(np.prod(cdims)) * self.dtype.itemsize
            if addr != UNDEF:
                for k, idx in enumerate(np.ndindex(*grid)):
                    self._place(out, f._read(addr + k * csize, csize), tuple(i * c for i, c in zip(idx, cdims)), cdims)
            return out
        if index == 3:   # fixed array: one entry per chunk of the grid, row-major
            addr = f._addr(b, off + 1)   # (one byte of page bits in front of the header address)
            grid = [-(-s // c) for s, c in zip(self.shape, cdims)]
            if addr != UNDEF:
                for idx, (caddr, size, mask) in zip(np.ndindex(*grid), f._fixed_array(addr, int(np.prod(grid)))):
                    if caddr == UNDEF:
                        continue
                    raw = f._read(caddr, size if size else int(np.prod(cdims)) * self.dtype.itemsize)
                    self._place(out, self._unfilter(raw, mask) if size else raw, tuple(i * c for i, c in zip(idx, cdims)), cdims)
            return out
        _unsupported({4: "extensible-array", 5: "v2 B-tree"}.get(index, f"type {index}") + " chunk index")


class H5File:
    """Read-only view of an HDF5 file: ``f.keys()``, ``f[name]`` -> numpy array (``h5py.File(path)[name][()]``)."""

    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self._superblock()
        self._links = self._group_links(self.root_header)

    # ---- raw access
    def _read(self, addr: int, n: int) -> bytes:
        a = self.base + addr
        if addr == UNDEF or a + n > len(self.buf):
            raise H5FormatError(f"address {addr:#x}+{n} beyond the end of the file ({len(self.buf)} bytes)")
        return self.buf[a: a + n]

    def _addr(self, b: bytes, off: int) -> int:
        v = int.from_bytes(b[off: off + self.O], "little")
        return UNDEF if v == (1 << (8 * self.O)) - 1 else v

    def _superblock(self):
        pos = 0
        while True:
            if self.buf[pos: pos + 8] == SIGNATURE:
                break
            pos = 512 if pos == 0 else pos * 2
            if pos + 8 > len(self.buf):
                raise H5FormatError("no HDF5 signature at 0, 512, 1024, ...: not an HDF5 file")
        b = self.buf
        ver = b[pos + 8]
        if ver in (0, 1):
            self.O, self.L = b[pos + 13], b[pos + 14]
            p = pos + 24 + (4 if ver == 1 else 0)
            self.base = 0
            base = self._addr(b, p)
            self.base = pos if base == 0 and pos else base   # a user block shifts every address by its length
            p += 4 * self.O                                   # base, free-space info, end of file, driver info
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root_header = self._addr(b, p + self.O)
            cache = struct.unpack_from("<I", b, p + 2 * self.O)[0]
            self._root_cache = (self._addr(b, p + 2 * self.O + 8), self._addr(b, p + 3 * self.O + 8)) if cache == 1 else None
        elif ver in (2, 3):
            self.O, self.L = b[pos + 9], b[pos + 10]
            self.base = 0
            base = self._addr(b, pos + 12)
            self.base = pos if base == 0 and pos else base
            self.root_header = self._addr(b, pos + 12 + 3 * self.O)
            self._root_cache = None
        else:
            raise H5FormatError(f"superblock version {ver}")
        if self.O not in (2, 4, 8) or self.L not in (2, 4, 8):
            raise H5FormatError(f"size of offsets / lengths {self.O} / {self.L}")

    # ---- object headers
    def _messages(self, addr: int) -> List[Tuple[int, int, bytes]]:
        head = self._read(addr, 16)
        out: List[Tuple[int, int, bytes]] = []
        if head[:4] == b"OHDR":
            if head[4] != 2:
                raise H5FormatError(f"object header version {head[4]}")
            flags = head[5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            w = 1 << (flags & 3)
            size = int.from_bytes(self._read(p, w), "little")
            p += w
            blocks = [(p, size)]
            order = 2 if flags & 0x04 else 0
            while blocks:
                p, size = blocks.pop(0)
                end = p + size
                while p + 4 + order <= end:
                    h = self._read(p, 4 + order)
                    mtype, msize, mflags = h[0], struct.unpack_from("<H", h, 1)[0], h[3]
                    p += 4 + order
                    if p + msize > end:   # gap before the checksum
                        break
                    body = self._read(p, msize)
                    p += msize
                    if mtype == 0x10:
                        caddr, clen = self._addr(body, 0), int.from_bytes(body[self.O: self.O + self.L], "little")
                        if self._read(caddr, 4) != b"OCHK":
                            raise H5FormatError("object header continuation without OCHK signature")
                        blocks.append((caddr + 4, clen - 8))   # signature in front, checksum behind
                    elif mtype != 0:
                        out.append((mtype, mflags, body))
            return out
        if head[0] != 1:
            raise H5FormatError(f"object header version {head[0]} at {addr:#x}")
        nmsgs = struct.unpack_from("<H", head, 2)[0]
        size = struct.unpack_from("<I", head, 8)[0]
        blocks = [(addr + 16, size)]
        seen = 0
        while blocks and seen < nmsgs:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and seen < nmsgs:
                h = self._read(p, 8)
                mtype, msize, mflags = struct.unpack_from("<HHB", h, 0)
                body = self._read(p + 8, msize)
                p += 8 + msize
                seen += 1
                if mtype == 0x10:
                    blocks.append((self._addr(body, 0), int.from_bytes(body[self.O: self.O + self.L], "little")))
                elif mtype != 0:
                    if mflags & 0x02:
                        _unsupported("shared header message")
                    out.append((mtype, mflags, body))
        return out

    # ---- groups
    def _heap_name(self, heap_data: int, off: int) -> str:
        a = self.base + heap_data + off
        e = self.buf.index(b"\0", a)
        return self.buf[a:e].decode("utf-8")

    def _symbol_table(self, btree: int, heap: int) -> Dict[str, int]:
        hd = self._read(heap, 8 + 2 * self.L + self.O)
        if hd[:4] != b"HEAP":
            raise H5FormatError("local heap signature")
        heap_data = self._addr(hd, 8 + 2 * self.L)
        links: Dict[str, int] = {}

        def walk(addr: int):
            nd = self._read(addr, 8 + 2 * self.O)
            if nd[:4] == b"SNOD":
                n = struct.unpack_from("<H", nd, 6)[0]
                esz = 2 * self.O + 24
                ent = self._read(addr + 8, n * esz)
                for i in range(n):
                    e = ent[i * esz: (i + 1) * esz]
                    links[self._heap_name(heap_data, self._addr(e, 0))] = self._addr(e, self.O)
                return
            if nd[:4] != b"TREE" or nd[4] != 0:
                raise H5FormatError("group B-tree node signature / type")
            n = struct.unpack_from("<H", nd, 6)[0]
            body = self._read(addr + 8 + 2 * self.O, (2 * n + 1) * max(self.L, self.O) + 8 * n)
            p = self.L
            for _ in range(n):
                walk(self._addr(body, p))
                p += self.O + self.L

        walk(btree)
        return links

    def _group_links(self, header: int) -> Dict[str, int]:
        links: Dict[str, int] = {}
        msgs = self._messages(header)
        for mtype, _fl, body in msgs:
            if mtype == 0x11:     # symbol table message
                links.update(self._symbol_table(self._addr(body, 0), self._addr(body, self.O)))
            elif mtype == 0x06:   # link message
                ver, fl = body[0], body[1]
                p = 2
                ltype = 0
                if fl & 0x08:
                    ltype = body[p]
                    p += 1
                if fl & 0x04:
                    p += 8
                if fl & 0x10:
                    p += 1
                w = 1 << (fl & 3)
                nlen = int.from_bytes(body[p: p + w], "little")
                p += w
                name = body[p: p + nlen].decode("utf-8")
                p += nlen
                if ltype == 0:
                    links[name] = self._addr(body, p)
            elif mtype == 0x02:   # link info: dense storage when a fractal heap is present
                fl = body[1]
                p = 2 + (8 if fl & 1 else 0)
                if self._addr(body, p) != UNDEF:
                    _unsupported("dense link storage (a group with more than 8 links written with libver='latest')")
        return links

    # ---- chunk index
    def _chunk_btree(self, addr: int, ndims: int):
        nd = self._read(addr, 8 + 2 * self.O)
        if nd[:4] != b"TREE" or nd[4] != 1:
            raise H5FormatError("chunk B-tree node signature / type")
        level, n = nd[5], struct.unpack_from("<H", nd, 6)[0]
        ksz = 8 + 8 * ndims
        body = self._read(addr + 8 + 2 * self.O, n * (ksz + self.O) + ksz)
        p = 0
        for _ in range(n):
            size, mask = struct.unpack_from("<II", body, p)
            offs = struct.unpack_from(f"<{ndims}Q", body, p + 8)
            child = self._addr(body, p + ksz)
            p += ksz + self.O
            if level == 0:
                yield size, mask, offs, child
            else:
                yield from self._chunk_btree(child, ndims)

    def _fixed_array(self, addr: int, nchunks: int):
        """(chunk address, stored size or 0 when unfiltered, filter mask) per chunk — "FAHD" header + "FADB" data block."""
        h = self._read(addr, 8 + self.L + self.O)
        if h[:4] != b"FAHD":
            raise H5FormatError("fixed-array header signature")
        client, esize, page_bits = h[5], h[6], h[7]
        nelm = int.from_bytes(h[8: 8 + self.L], "little")
        dblk = self._addr(h, 8 + self.L)
        if nelm < nchunks:
            raise H5FormatError("fixed array holds fewer entries than the chunk grid")
        if dblk == UNDEF:
            return [(UNDEF, 0, 0)] * nchunks
        if self._read(dblk, 4) != b"FADB":
            raise H5FormatError("fixed-array data block signature")
        p = dblk + 6 + self.O
        page = 1 << page_bits
        def entries(at: int, n: int):
            raw = self._read(at, n * esize)
            for i in range(n):
                e = raw[i * esize: (i + 1) * esize]
                a = self._addr(e, 0)
                if client == 0:
                    yield a, 0, 0
                else:
                    w = esize - self.O - 4
                    yield a, int.from_bytes(e[self.O: self.O + w], "little"), struct.unpack_from("<I", e, self.O + w)[0]
        if nelm <= page:
            return list(entries(p, nchunks))
        npages = -(-nelm // page)
        bitmap = self._read(p, (npages + 7) // 8)
        p += (npages + 7) // 8 + 4          # the data block's own checksum follows the bitmap; pages follow it
        out = []
        for pg in range(npages):
            n = min(page, nelm - pg * page)
            if bitmap[pg // 8] >> (7 - pg % 8) & 1:
                out.extend(entries(p, n))
            else:
                out.extend([(UNDEF, 0, 0)] * n)   # page never written
            p += n * esize + 4
        return out[:nchunks]

    # ---- mapping interface
    def keys(self) -> List[str]:
        return sorted(self._links)

    def __contains__(self, name: str) -> bool:
        return name.strip("/") in self._links or self._resolve(name, missing_ok=True) is not None

    def _resolve(self, name: str, missing_ok: bool = False) -> Optional[int]:
        links = self._links
        parts = [p for p in name.split("/") if p]
        addr = None
        for i, part in enumerate(parts):
            if part not in links:
                if missing_ok:
                    return None
                raise KeyError(f"'{name}' (have: {', '.join(sorted(links))})")
            addr = links[part]
            if i + 1 < len(parts):
                links = self._group_links(addr)
        return addr

    def __getitem__(self, name: str) -> np.ndarray:
        addr = self._resolve(name)
        return _Dataset(self, name, self._messages(addr)).read()


def read_h5(path: str, keys: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
    """Every dataset of the root group (or the named ones) as numpy arrays."""
    f = H5File(path)
    return {k: f[k] for k in (keys if keys is not None else f.keys())}


class LoadH5:
    """``LoadH5(path_key, keys)`` of training_project/utils/my_transform.py:142-154: ``d[key] = file[key][()]`` for every key,
    the path entry stays in the dictionary."""

    def __init__(self, path_key: str, keys):
        self.path_key = path_key
        self.keys = (keys,) if isinstance(keys, str) else tuple(keys)

    def __call__(self, data):
        d = dict(data)
        f = H5File(d[self.path_key])
        for key in self.keys:
            d[key] = f[key]
        return d


# ===================================================================================================== writer
def _pad8(b: bytes) -> bytes:
    return b + bytes(-len(b) % 8)


def _dtype_message(dt: np.dtype) -> bytes:
    size = dt.itemsize
    big = dt.byteorder == ">"
    if dt.kind in "iu":
        bits = (1 if big else 0) | (0x08 if dt.kind == "i" else 0)
        return struct.pack("<BBBBI", 0x10, bits, 0, 0, size) + struct.pack("<HH", 0, 8 * size)
    if dt.kind == "f" and size in (2, 4, 8):
        eloc, esize, msize, bias = {2: (10, 5, 10, 15), 4: (23, 8, 23, 127), 8: (52, 11, 52, 1023)}[size]
        return struct.pack("<BBBBI", 0x11, 0x20 | (1 if big else 0), 8 * size - 1, 0, size) + \
            struct.pack("<HHBBBBI", 0, 8 * size, eloc, esize, 0, msize, bias)
    raise TypeError(f"write_h5: dtype {dt} (integers and IEEE floats only)")


def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(msgs: List[bytes]) -> bytes:
    blob = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(blob)) + blob


def write_h5(path: str, arrays: Dict[str, np.ndarray]) -> None:
    """One contiguous dataset per entry in the root group, laid out as ``h5py.File(path, 'w')`` + ``f[key] = array`` does
    (preprocess/to_h5.py:40-50): superblock v0, root symbol table (B-tree v1, local heap, one symbol node)."""
    names = sorted(arrays, key=lambda s: s.encode("utf-8"))
    if not names:
        raise ValueError("write_h5: nothing to write")
    for n in names:
        if not n or "/" in n or "\0" in n:
            raise ValueError(f"write_h5: dataset name {n!r} (root-level names only)")
    leaf_k = max(4, (len(names) + 1) // 2)          # a symbol node holds 2K entries: all of them fit into one
    arrs = {n: np.ascontiguousarray(arrays[n]) for n in names}

    # local heap data: the empty string at 0, then the names, each 8-byte aligned
    heap = bytearray(8)
    name_off = {}
    for n in names:
        name_off[n] = len(heap)
        heap += _pad8(n.encode("utf-8") + b"\0")
    heap_data = bytes(heap)

    # fixed-size pieces first: addresses are assigned front to back
    sb_size = 56 + 40
    root_msgs_size = 16 + 8 + 16                     # header prefix + message prefix + symbol table message body
    pos = sb_size
    root_addr = pos
    pos += root_msgs_size
    btree_addr = pos
    btree_size = 8 + 16 + (2 * 2 * 16 + 1) * 8       # internal K = 16: room for 2K children + 2K+1 keys of 8 bytes
    pos += btree_size
    heap_addr = pos
    pos += 8 + 16 + 8
    heap_data_addr = pos
    pos += len(heap_data)
    snod_addr = pos
    snod_size = 8 + 2 * leaf_k * 40
    pos += snod_size

    headers, data_addr, hdr_addr = {}, {}, {}
    for n in names:
        a = arrs[n]
        space = struct.pack("<BBB5x", 1, a.ndim, 0) + b"".join(struct.pack("<Q", s) for s in a.shape)
        fill = struct.pack("<BBBB", 2, 2, 2, 0)      # version 2: allocate late, write the fill value if set, none defined
        layout = struct.pack("<BBQQ", 3, 1, 0, a.nbytes)   # address patched below
        headers[n] = [space, _dtype_message(a.dtype), fill, layout]
        hdr_addr[n] = pos
        pos += len(_object_header([_message(0x01, space), _message(0x03, headers[n][1], 1), _message(0x05, fill),
                                   _message(0x08, layout)]))
    for n in names:
        pos = (pos + 7) // 8 * 8
        data_addr[n] = pos if arrs[n].nbytes else UNDEF
        pos += arrs[n].nbytes
    eof = pos

    out = bytearray()
    # superblock v0
    out += SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", leaf_k, 16, 0)
    out += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    out += struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr)   # root symbol table entry
    assert len(out) == sb_size
    out += _object_header([_message(0x11, struct.pack("<QQ", btree_addr, heap_addr))])
    assert len(out) == btree_addr
    # B-tree: one leaf-level child; key 0 = the empty string, key 1 = the greatest name of the child
    node = b"TREE" + bytes([0, 0]) + struct.pack("<H", 1) + struct.pack("<QQ", UNDEF, UNDEF)
    node += struct.pack("<QQQ", 0, snod_addr, name_off[names[-1]])
    out += node + bytes(btree_size - len(node))
    out += b"HEAP" + bytes(4) + struct.pack("<QQQ", len(heap_data), 1, heap_data_addr)   # free-list head 1 = none
    out += heap_data
    snod = b"SNOD" + bytes([1, 0]) + struct.pack("<H", len(names))
    for n in names:
        snod += struct.pack("<QQII16x", name_off[n], hdr_addr[n], 0, 0)
    out += snod + bytes(snod_size - len(snod))
    for n in names:
        space, dtm, fill, _ = headers[n]
        layout = struct.pack("<BBQQ", 3, 1, data_addr[n], arrs[n].nbytes)
        assert len(out) == hdr_addr[n]
        out += _object_header([_message(0x01, space), _message(0x03, dtm, 1), _message(0x05, fill), _message(0x08, layout)])
    for n in names:
        out += bytes(-len(out) % 8)
        out += arrs[n].tobytes()
    assert len(out) == eof
    with open(path, "wb") as fh:
        fh.write(bytes(out))
