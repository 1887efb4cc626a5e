"""Minimal pure-Python HDF5 reader / writer -- just enough of the file format for Keras weight files.

The reference stores weights with Keras 2.3.1 -> h5py -> libhdf5 (`model.save_weights` / `model.save` /
`ModelCheckpoint`: /root/reference/unet/unet.py:261-264,361-379, vae/lattice_vae.py:149-151,339-341; the
published U-Net is an LFS `.h5`, /root/reference/.gitattributes:1-2).  h5py is not installable here, so
this module implements the subset of the HDF5 File Format Specification (version 1.1 / 2.0 structures)
those files use:

  read : superblock v0/v1 (and v2/v3 as far as the root address), old-style groups (symbol-table message
         -> v1 B-tree of SNOD nodes + local heap), v1 object headers with continuation blocks, attribute
         messages v1-v3, dataspace v1/v2, datatypes fixed/float/string/vlen-string, data layout v3
         contiguous / compact / chunked (v1 chunk B-tree, no filters), global heap (vlen strings).
  write: superblock v0, old-style groups with one SNOD per group (the group-leaf K in the superblock is
         sized to the largest group), v1 object headers, v1 attributes (fixed-length strings, float/int
         arrays), contiguous little-endian float32/float64/int32/int64 datasets.

Files written here are valid HDF5: tests/test_hdf5_min.py reads them back with the real libhdf5 (h5dump /
ctypes) when the image has it, and reads libhdf5-written Keras-layout fixtures (tests/golden/*.h5) with
this reader.  New-style groups (object header v2 / fractal heaps, libver="latest") are rejected loudly.
"""
from __future__ import annotations

import os
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


# ============================================================================================ reader
class _Datatype:
    def __init__(self, cls, size, np_dtype=None, strpad=None, vlen_base=None, vlen_string=False):
        self.cls, self.size, self.np_dtype = cls, size, np_dtype
        self.strpad, self.vlen_base, self.vlen_string = strpad, vlen_base, vlen_string


def _parse_datatype(buf, o=0):
    cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, o)
    cls, ver = cv & 0x0F, cv >> 4
    if ver not in (1, 2, 3):
        raise Hdf5Error("datatype version %d" % ver)
    if cls == 0:      # fixed point
        order = ">" if (b0 & 1) else "<"
        signed = bool(b0 & 8)
        return _Datatype(0, size, np.dtype("%s%s%d" % (order, "i" if signed else "u", size)))
    if cls == 1:      # floating point (IEEE assumed for 2/4/8 bytes)
        order = ">" if (b0 & 1) else "<"
        if size not in (2, 4, 8):
            raise Hdf5Error("float size %d" % size)
        return _Datatype(1, size, np.dtype("%sf%d" % (order, size)))
    if cls == 3:      # fixed-length string
        return _Datatype(3, size, np.dtype("S%d" % size), strpad=b0 & 0x0F)
    if cls == 9:      # variable length: sequence or string
        base = _parse_datatype(buf, o + 8)
        return _Datatype(9, size, None, vlen_base=base, vlen_string=(b0 & 0x0F) == 1)
    return _Datatype(cls, size, None)


def _parse_dataspace(buf, o=0):
    ver, rank, flags = struct.unpack_from("<BBB", buf, o)
    if ver == 1:
        p = o + 8
    elif ver == 2:
        if buf[o + 3] == 2:          # null dataspace
            return None
        p = o + 4
    else:
        raise Hdf5Error("dataspace version %d" % ver)
    return tuple(struct.unpack_from("<%dQ" % rank, buf, p)) if rank else ()


class Hdf5Object:
    """A group or a dataset.  Groups: `.keys()`, `[name]` (paths with '/' allowed), `in`; datasets:
    `.shape`, `.dtype`, `.read()` / `[()]`; both: `.attrs` (dict name -> numpy array / bytes / scalar)."""

    def __init__(self, f, addr, name="/"):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = f._read_object_header(addr)
        self._links = None
        self._attrs = None

    # ---- attributes
    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = {}
            for t, body in self._msgs:
                if t == 0x000C:
                    k, v = self._f._parse_attribute(body)
                    self._attrs[k] = v
        return self._attrs

    # ---- group interface
    @property
    def is_group(self):
        return any(t == 0x0011 for t, _ in self._msgs)

    def _load_links(self):
        if self._links is None:
            st = [b for t, b in self._msgs if t == 0x0011]
            if not st:
                if any(t in (0x0002, 0x0006) for t, _ in self._msgs):
                    raise Hdf5Error("new-style group (link messages): not supported; re-save with libver='earliest'")
                self._links = {}
            else:
                btree, heap = struct.unpack_from("<QQ", st[0], 0)
                self._links = self._f._read_group(btree, heap)
        return self._links

    def keys(self):
        return list(self._load_links().keys())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        if name == ():
            return self.read()
        obj = self
        for part in [p for p in name.split("/") if p]:
            links = obj._load_links()
            if part not in links:
                raise KeyError(name)
            obj = Hdf5Object(self._f, links[part], (obj.name.rstrip("/") + "/" + part))
        return obj

    # ---- dataset interface
    @property
    def is_dataset(self):
        return any(t == 0x0008 for t, _ in self._msgs)

    def _dataset_meta(self):
        dt = sp = lay = None
        for t, b in self._msgs:
            if t == 0x0003:
                dt = _parse_datatype(b)
            elif t == 0x0001:
                sp = _parse_dataspace(b)
            elif t == 0x0008:
                lay = b
            elif t == 0x000B and len(b) >= 2 and b[1] > 0:
                raise Hdf5Error("%s: filtered (compressed) datasets are not supported" % self.name)
        if dt is None or lay is None:
            raise Hdf5Error("%s is not a dataset" % self.name)
        return dt, sp, lay

    @property
    def shape(self):
        return self._dataset_meta()[1]

    @property
    def dtype(self):
        return self._dataset_meta()[0].np_dtype

    def read(self):
        dt, shape, lay = self._dataset_meta()
        if dt.np_dtype is None:
            raise Hdf5Error("%s: unsupported dataset datatype class %d" % (self.name, dt.cls))
        shape = () if shape is None else shape
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        nbytes = n * dt.size
        ver = lay[0]
        if ver != 3:
            raise Hdf5Error("data layout version %d" % ver)
        cls = lay[1]
        if cls == 0:                                   # compact
            size = struct.unpack_from("<H", lay, 2)[0]
            raw = bytes(lay[4:4 + size])
        elif cls == 1:                                 # contiguous
            addr, size = struct.unpack_from("<QQ", lay, 2)
            raw = b"\0" * nbytes if addr == UNDEF else self._f._read(addr, nbytes)
        elif cls == 2:                                 # chunked, v1 B-tree
            rank = lay[2]
            btree = struct.unpack_from("<Q", lay, 3)[0]
            cdims = struct.unpack_from("<%dI" % rank, lay, 11)
            return self._f._read_chunked(btree, shape, cdims[:-1], dt)
        else:
            raise Hdf5Error("data layout class %d" % cls)
        a = np.frombuffer(raw, dtype=dt.np_dtype, count=n).reshape(shape)
        return a.copy()


class Hdf5File(Hdf5Object):
    """Read-only HDF5 file: `Hdf5File(path)["model_weights/conv3d_1/conv3d_1/kernel:0"].read()`."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self._buf = fh.read()
        base = self._buf.find(SIG)
        if base != 0:
            raise Hdf5Error("%s is not an HDF5 file (no signature at offset 0)" % path)
        ver = self._buf[8]
        if ver in (0, 1):
            so, sl = self._buf[13], self._buf[14]
            if (so, sl) != (8, 8):
                raise Hdf5Error("only 8-byte offsets/lengths are supported")
            p = 24 if ver == 0 else 28
            self._base = struct.unpack_from("<Q", self._buf, p)[0]
            root_entry = p + 32
            root_addr = struct.unpack_from("<Q", self._buf, root_entry + 8)[0]
        elif ver in (2, 3):
            if (self._buf[9], self._buf[10]) != (8, 8):
                raise Hdf5Error("only 8-byte offsets/lengths are supported")
            self._base, _ext, _eof, root_addr = struct.unpack_from("<QQQQ", self._buf, 12)
        else:
            raise Hdf5Error("superblock version %d" % ver)
        super().__init__(self, root_addr, "/")

    def close(self):
        self._buf = b""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- raw access
    def _read(self, addr, n):
        a = self._base + addr
        if a + n > len(self._buf):
            raise Hdf5Error("read past end of file (truncated?)")
        return self._buf[a:a + n]

    # ---- object headers
    def _read_object_header(self, addr):
        head = self._read(addr, 16)
        if head[:4] == b"OHDR":
            raise Hdf5Error("version-2 object headers (libver='latest') are not supported")
        ver, _, nmsg, _refs, hsize = struct.unpack_from("<BBHII", head, 0)
        if ver != 1:
            raise Hdf5Error("object header version %d at %#x" % (ver, addr))
        msgs = []
        blocks = [(addr + 16, hsize)]
        while blocks and len(msgs) < nmsg:
            baddr, blen = blocks.pop(0)
            blk = self._read(baddr, blen)
            o = 0
            while o + 8 <= blen and len(msgs) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", blk, o)
                body = blk[o + 8:o + 8 + msize]
                o += 8 + msize
                msgs.append((mtype, body))
                if mtype == 0x0010:
                    coff, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((coff, clen))
        return msgs

    # ---- old-style groups
    def _heap_string(self, heap_data, off):
        end = heap_data.index(b"\0", off)
        return heap_data[off:end].decode("utf-8")

    def _read_group(self, btree_addr, heap_addr):
        h = self._read(heap_addr, 32)
        if h[:4] != b"HEAP":
            raise Hdf5Error("bad local heap signature")
        dsize, _free, daddr = struct.unpack_from("<QQQ", h, 8)
        heap = self._read(daddr, dsize)
        links = {}
        self._walk_group_btree(btree_addr, heap, links)
        return links

    def _walk_group_btree(self, addr, heap, links):
        if addr == UNDEF:
            return
        head = self._read(addr, 24)
        if head[:4] == b"SNOD":
            self._read_snod(addr, heap, links)
            return
        if head[:4] != b"TREE":
            raise Hdf5Error("bad B-tree signature at %#x" % addr)
        ntype, level, used = struct.unpack_from("<BBH", head, 4)
        if ntype != 0:
            raise Hdf5Error("expected a group B-tree node")
        body = self._read(addr + 24, (2 * used + 1) * 8)
        for i in range(used):
            child = struct.unpack_from("<Q", body, 8 + 16 * i)[0]
            if level > 0:
                self._walk_group_btree(child, heap, links)
            else:
                self._read_snod(child, heap, links)

    def _read_snod(self, addr, heap, links):
        head = self._read(addr, 8)
        if head[:4] != b"SNOD":
            raise Hdf5Error("bad symbol table node signature at %#x" % addr)
        nsym = struct.unpack_from("<H", head, 6)[0]
        ents = self._read(addr + 8, 40 * nsym)
        for i in range(nsym):
            noff, oaddr = struct.unpack_from("<QQ", ents, 40 * i)
            links[self._heap_string(heap, noff)] = oaddr

    # ---- attributes
    def _parse_attribute(self, b):
        ver = b[0]
        if ver == 1:
            nsz, tsz, ssz = struct.unpack_from("<HHH", b, 2)
            pad = lambda v: (v + 7) & ~7
            o = 8
            name = b[o:o + nsz].split(b"\0")[0].decode("utf-8"); o += pad(nsz)
            dt = _parse_datatype(b, o); o += pad(tsz)
            shape = _parse_dataspace(b, o); o += pad(ssz)
        elif ver in (2, 3):
            nsz, tsz, ssz = struct.unpack_from("<HHH", b, 2)
            o = 8 + (1 if ver == 3 else 0)
            name = b[o:o + nsz].split(b"\0")[0].decode("utf-8"); o += nsz
            dt = _parse_datatype(b, o); o += tsz
            shape = _parse_dataspace(b, o); o += ssz
        else:
            raise Hdf5Error("attribute message version %d" % ver)
        return name, self._decode_values(b, o, dt, shape)

    def _decode_values(self, b, o, dt, shape):
        if shape is None:
            return None
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if dt.cls == 9 and dt.vlen_string:
            out = []
            for i in range(n):
                ln, gaddr, gidx = struct.unpack_from("<IQI", b, o + 16 * i)
                out.append(self._global_heap_object(gaddr, gidx)[:ln] if ln else b"")
            arr = np.array(out, dtype=object).reshape(shape) if shape else out[0]
            return arr
        if dt.np_dtype is None:
            return None
        a = np.frombuffer(b, dtype=dt.np_dtype, count=n, offset=o).reshape(shape).copy()
        if dt.cls == 3 and dt.strpad == 2:              # space padded
            a = np.char.rstrip(a, b" ")
        return a if shape else a[()]

    def _global_heap_object(self, addr, idx):
        head = self._read(addr, 16)
        if head[:4] != b"GCOL":
            raise Hdf5Error("bad global heap signature")
        csize = struct.unpack_from("<Q", head, 8)[0]
        blk = self._read(addr, csize)
        o = 16
        while o + 16 <= csize:
            oi, _rc, _r, osz = struct.unpack_from("<HHIQ", blk, o)
            if oi == 0:
                break
            if oi == idx:
                return blk[o + 16:o + 16 + osz]
            o += 16 + ((osz + 7) & ~7)
        raise Hdf5Error("global heap object %d not found" % idx)

    # ---- chunked datasets (v1 B-tree, node type 1, no filters)
    def _read_chunked(self, btree, shape, cdims, dt):
        out = np.zeros(shape, dtype=dt.np_dtype)
        rank = len(shape)
        csize = int(np.prod(cdims, dtype=np.int64)) * dt.size

        def walk(addr):
            head = self._read(addr, 24)
            if head[:4] != b"TREE":
                raise Hdf5Error("bad chunk B-tree signature")
            ntype, level, used = struct.unpack_from("<BBH", head, 4)
            if ntype != 1:
                raise Hdf5Error("expected a chunk B-tree node")
            ksz = 8 + 8 * (rank + 1)
            body = self._read(addr + 24, used * (ksz + 8) + ksz)
            for i in range(used):
                ko = i * (ksz + 8)
                nbytes, fmask = struct.unpack_from("<II", body, ko)
                offs = struct.unpack_from("<%dQ" % (rank + 1), body, ko + 8)[:rank]
                child = struct.unpack_from("<Q", body, ko + ksz)[0]
                if level > 0:
                    walk(child)
                    continue
                if fmask != 0 or nbytes != csize:
                    raise Hdf5Error("filtered chunks are not supported")
                chunk = np.frombuffer(self._read(child, csize), dtype=dt.np_dtype).reshape(cdims)
                sl = tuple(slice(o_, min(o_ + c, s)) for o_, c, s in zip(offs, cdims, shape))
                out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]

        if btree != UNDEF:
            walk(btree)
        return out


# ============================================================================================ writer
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _dt_bytes(dtype):
    dtype = np.dtype(dtype)
    if dtype.kind == "f" and dtype.itemsize == 4:
        return struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    if dtype.kind == "f" and dtype.itemsize == 8:
        return struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    if dtype.kind in "iu" and dtype.itemsize in (1, 2, 4, 8):
        return struct.pack("<BBBBI", 0x10, 0x08 if dtype.kind == "i" else 0, 0, 0, dtype.itemsize) + \
            struct.pack("<HH", 0, 8 * dtype.itemsize)
    if dtype.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dtype.itemsize)     # null-padded ASCII, like numpy 'S'
    raise Hdf5Error("cannot store dtype %s" % dtype)


def _space_bytes(shape):
    if shape == ():
        return struct.pack("<BBBB4x", 1, 0, 0, 0)
    return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + struct.pack("<%dQ" % len(shape), *shape)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attr_msg(name, value):
    if isinstance(value, str):
        value = value.encode("utf-8")
    if isinstance(value, (bytes, bytearray)):
        value = np.array(bytes(value), dtype="S%d" % max(len(value), 1))
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype.kind == "S" and a.dtype.itemsize == 0:
        a = a.astype("S1")
    if a.dtype.kind == "f" and a.dtype.itemsize not in (4, 8):
        a = a.astype(np.float32)
    a = np.ascontiguousarray(a)
    nm = name.encode("utf-8") + b"\0"
    dt, sp = _dt_bytes(a.dtype), _space_bytes(a.shape)
    body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp) + a.tobytes()
    if len(body) > 65000:
        raise Hdf5Error("attribute %s too large for one object-header message (split it, as Keras does)" % name)
    return _msg(0x000C, body)


class Hdf5Writer:
    """Build a tree, then write():

        w = Hdf5Writer()
        g = w.root.create_group("conv3d_1"); g.attrs["weight_names"] = np.array([b"conv3d_1/kernel:0"])
        g.create_dataset("conv3d_1/kernel:0", array)       # intermediate groups are created
        w.write(path)
    """

    class _Node:
        def __init__(self):
            self.children = {}     # name -> _Node (insertion order kept; written sorted, as the format requires)
            self.attrs = {}
            self.data = None

        def create_group(self, name):
            node = self
            for part in [p for p in name.split("/") if p]:
                if part not in node.children:
                    node.children[part] = Hdf5Writer._Node()
                node = node.children[part]
                if node.data is not None:
                    raise Hdf5Error("%s is a dataset" % part)
            return node

        def create_dataset(self, name, data):
            parts = [p for p in name.split("/") if p]
            parent = self.create_group("/".join(parts[:-1])) if len(parts) > 1 else self
            a = np.asarray(data)
            if a.dtype.kind == "f" and a.dtype.itemsize not in (4, 8):
                a = a.astype(np.float32)
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            node = Hdf5Writer._Node()
            node.data = np.ascontiguousarray(a)
            parent.children[parts[-1]] = node
            return node

        def __getitem__(self, name):
            node = self
            for part in [p for p in name.split("/") if p]:
                node = node.children[part]
            return node

    def __init__(self):
        self.root = Hdf5Writer._Node()

    def write(self, path):
        def max_children(node):
            return max([len(node.children)] + [max_children(c) for c in node.children.values()])

        leaf_k = max(4, (max_children(self.root) + 1) // 2)     # one SNOD (2K entries) holds any group
        internal_k = 16
        out = bytearray(b"\0" * 96)                              # superblock placeholder

        def alloc(b):
            while len(out) % 8:
                out.append(0)
            addr = len(out)
            out.extend(b)
            return addr

        def header(msgs):
            body = b"".join(msgs)
            return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body

        def write_node(node):
            """returns (object header address, btree address | None, heap address | None)"""
            attr_msgs = [_attr_msg(k, v) for k, v in node.attrs.items()]
            if node.data is not None:
                a = node.data
                daddr = alloc(a.tobytes()) if a.size else UNDEF
                msgs = [_msg(0x0001, _space_bytes(a.shape)), _msg(0x0003, _dt_bytes(a.dtype), 1),
                        _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0)),          # fill value v2: late alloc, undefined
                        _msg(0x0008, struct.pack("<BBQQ", 3, 1, daddr, a.nbytes))] + attr_msgs
                return alloc(header(msgs)), None, None
            kids = []
            for name in sorted(node.children, key=lambda s: s.encode("utf-8")):
                kids.append((name, write_node(node.children[name])))
            # local heap: offset 0 = "", then the names, 8-aligned
            heap = bytearray(b"\0" * 8)
            offs = []
            for name, _ in kids:
                offs.append(len(heap))
                heap.extend(_pad8(name.encode("utf-8") + b"\0"))
            # keep a properly described free block at the end (libhdf5 wants >= 16 bytes to describe it)
            free_off = len(heap)
            heap.extend(struct.pack("<QQ", 1, 16))                # next = H5HL_FREE_NULL (1), size 16
            heap_data = alloc(bytes(heap))
            heap_addr = alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, heap_data))
            # one symbol-table node
            snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(kids)))
            for (name, (oaddr, bt, hp)), noff in zip(kids, offs):
                if bt is not None:
                    snod.extend(struct.pack("<QQII", noff, oaddr, 1, 0) + struct.pack("<QQ", bt, hp))
                else:
                    snod.extend(struct.pack("<QQII", noff, oaddr, 0, 0) + b"\0" * 16)
            snod.extend(b"\0" * (8 + 40 * 2 * leaf_k - len(snod)))
            snod_addr = alloc(bytes(snod))
            # B-tree node (level 0) with that single child; full allocated size (2K+1 keys, 2K children)
            bt = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if kids else 0, UNDEF, UNDEF))
            if kids:
                bt.extend(struct.pack("<QQQ", 0, snod_addr, offs[-1]))
            bt.extend(b"\0" * (24 + (2 * internal_k + 1) * 8 + 2 * internal_k * 8 - len(bt)))
            bt_addr = alloc(bytes(bt))
            msgs = [_msg(0x0011, struct.pack("<QQ", bt_addr, heap_addr))] + attr_msgs
            return alloc(header(msgs)), bt_addr, heap_addr

        root_addr, bt, hp = write_node(self.root)
        while len(out) % 8:
            out.append(0)
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, leaf_k, internal_k, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(out), UNDEF)
        sb += struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", bt, hp)
        assert len(sb) == 96
        out[0:96] = sb
        # never truncate the target in place: a kill during the write would destroy the only copy of the best weights
        tmp = "%s.tmp.%d" % (path, os.getpid())
        try:
            with open(tmp, "wb") as fh:
                fh.write(bytes(out))
                fh.flush()
                os.fsync(fh.fileno())
            os.replace(tmp, path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)


def is_hdf5(path):
    try:
        with open(path, "rb") as fh:
            return fh.read(8) == SIG
    except OSError:
        return False
