"""A small pure-Python reader AND writer for the subset of HDF5 that Keras weight files use (no h5py / libhdf5 in this image's
test interpreter).

The reference loads pretrained ViT weights with `model.load_weights(path_to.h5)` (models/backbones/vision_transformer.py:149-169);
Keras writes such files through h5py with the library defaults, i.e. the "earliest" file-format features:
  superblock version 0 or 1, object headers version 1, groups as symbol tables (B-tree v1 + local heap + SNOD nodes),
  datasets with contiguous (or compact) layout and no filters, fixed-point / IEEE-float / fixed-length-string datatypes,
  attributes as version-1..3 attribute messages (`layer_names`, `weight_names`: arrays of byte strings - fixed-length with h5py 2,
  variable-length (global heap) with h5py 3; `backend` / `keras_version`: variable-length scalars).
Exactly that subset is implemented, from the HDF5 File Format Specification (version 2.0); anything else (chunked or filtered
datasets, version-2 object headers / dense groups, variable-length data) raises NotImplementedError naming the feature.
Validated against files written by h5py 3.3 / libhdf5 1.10.6 (tests/golden/keras_weights_*.h5, made by tests/golden/make_h5_golden.py).
The writer (`save_keras_weights`, bottom of the file) emits the same "earliest" structures - superblock 0, version-1 object headers in
one block, symbol-table groups (local heap, SNOD leaves of 8, B-tree nodes of 32, as many levels as the links need), contiguous
datasets, fixed-length string attributes - and is checked the other way round: h5py / libhdf5 reads what it wrote
(tests/golden/check_h5_with_h5py.py, run by tests/test_hdf5_lite.py in the image's conda interpreter).
"""
import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(ValueError):
    pass


class _Buf:
    def __init__(self, data):
        self.d = data

    def u(self, off, n):
        return int.from_bytes(self.d[off:off + n], "little")


class File:
    """f = File(path); f.attrs -> dict; f.keys(); f[name] -> Group | Dataset."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = _Buf(fh.read())
        d = self.buf.d
        base = None
        for off in (0, 512, 1024, 2048, 4096):
            if d[off:off + 8] == _SIG:
                base = off
                break
        if base is None:
            raise Hdf5Error("not an HDF5 file (no signature)")
        ver = d[base + 8]
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (files written with libver='latest'); Keras / h5py defaults write version 0" % ver)
        self.so, self.sl = d[base + 13], d[base + 14]            # size of offsets / lengths
        if self.so != 8 or self.sl != 8:
            raise NotImplementedError("HDF5 offsets / lengths of %d / %d bytes" % (self.so, self.sl))
        p = base + 24 if ver == 0 else base + 28                  # v1 adds indexed-storage K + reserved
        self.base_addr = self.buf.u(p, 8)
        p += 32                                                   # base, free-space, end-of-file, driver-info addresses
        # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self.root = Group(self, self.buf.u(p + 8, 8) + self.base_addr, "/")

    # ---- low level -------------------------------------------------------------------------------------------------
    def _messages(self, addr):
        """(type, flags, payload offset, payload size) of every message of a version-1 object header (continuations followed)."""
        b = self.buf
        if b.d[addr:addr + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers (libver='latest')")
        if b.d[addr] != 1:
            raise Hdf5Error("object header version %d at %#x" % (b.d[addr], addr))
        nmsg = b.u(addr + 2, 2)
        size = b.u(addr + 8, 4)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = b.u(p, 2), b.u(p + 2, 2), b.d[p + 4]
                body = p + 8
                if mtype == 0x0010:                              # continuation
                    blocks.append((b.u(body, 8) + self.base_addr, b.u(body + 8, 8)))
                out.append((mtype, flags, body, msize))
                p = body + msize
        return out

    @property
    def attrs(self):
        return self.root.attrs

    def keys(self):
        return self.root.keys()

    def __getitem__(self, name):
        return self.root[name]

    def __contains__(self, name):
        return name in self.root


def _datatype(buf, p):
    """-> (numpy dtype or ('S', n), size in bytes)."""
    cls_ver = buf.d[p]
    cls, ver = cls_ver & 0x0F, cls_ver >> 4
    bits0 = buf.d[p + 1]
    size = buf.u(p + 4, 4)
    if ver not in (1, 2, 3):
        raise NotImplementedError("datatype message version %d" % ver)
    order = ">" if (bits0 & 1) else "<"
    if cls == 0:                                                 # fixed point
        signed = bool(bits0 & 0x08)
        return np.dtype("%s%s%d" % (order, "i" if signed else "u", size)), size
    if cls == 1:                                                 # floating point: IEEE layouts only
        if size not in (2, 4, 8):
            raise NotImplementedError("%d-byte floating point" % size)
        return np.dtype("%sf%d" % (order, size)), size
    if cls == 3:                                                 # fixed-length string
        return ("S", size), size
    if cls == 9:                                                 # variable length: strings only (h5py >= 3 stores lists of bytes so)
        if (bits0 & 0x0F) != 1:
            raise NotImplementedError("variable-length sequences (only variable-length strings are read)")
        return ("V", size), size
    raise NotImplementedError("HDF5 datatype class %d" % cls)


def _dataspace(buf, p):
    ver = buf.d[p]
    rank = buf.d[p + 1]
    flags = buf.d[p + 2]
    if ver == 1:
        q = p + 8
    elif ver == 2:
        if buf.d[p + 3] == 2:
            return None                                          # null dataspace
        q = p + 4
    else:
        raise NotImplementedError("dataspace message version %d" % ver)
    dims = tuple(buf.u(q + 8 * i, 8) for i in range(rank))
    del flags
    return dims


def _pad8(n):
    return (n + 7) & ~7


class _Node:
    def __init__(self, f, addr, name):
        self.file, self.addr, self.name = f, addr, name
        self._msgs = f._messages(addr)

    @property
    def attrs(self):
        out = {}
        buf = self.file.buf
        for mtype, _flags, p, _size in self._msgs:
            if mtype != 0x000C:
                continue
            ver = buf.d[p]
            if ver == 1:
                nsz, tsz, ssz = buf.u(p + 2, 2), buf.u(p + 4, 2), buf.u(p + 6, 2)
                q = p + 8
                name = bytes(buf.d[q:q + nsz]).split(b"\x00")[0].decode()
                q += _pad8(nsz)
                dt, esz = _datatype(buf, q)
                q += _pad8(tsz)
                dims = _dataspace(buf, q)
                q += _pad8(ssz)
            elif ver in (2, 3):
                nsz, tsz, ssz = buf.u(p + 2, 2), buf.u(p + 4, 2), buf.u(p + 6, 2)
                q = p + 8 + (1 if ver == 3 else 0)               # v3: name character-set byte
                name = bytes(buf.d[q:q + nsz]).split(b"\x00")[0].decode()
                q += nsz
                dt, esz = _datatype(buf, q)
                q += tsz
                dims = _dataspace(buf, q)
                q += ssz
            else:
                raise NotImplementedError("attribute message version %d" % ver)
            out[name] = _decode(buf.d, q, dt, esz, dims, self.file.base_addr)
        if any(m[0] == 0x0015 for m in self._msgs):
            raise NotImplementedError("densely stored attributes (more than the object header holds)")
        return out


def _global_heap_object(data, base_addr, addr, index):
    """Object `index` of the global heap collection at `addr` (variable-length data lives there)."""
    a = addr + base_addr
    if bytes(data[a:a + 4]) != b"GCOL":
        raise Hdf5Error("bad global heap collection at %#x" % a)
    size = int.from_bytes(data[a + 8:a + 16], "little")
    p, end = a + 16, a + size
    while p + 16 <= end:
        idx = int.from_bytes(data[p:p + 2], "little")
        osz = int.from_bytes(data[p + 8:p + 16], "little")
        if idx == index:
            return bytes(data[p + 16:p + 16 + osz])
        if idx == 0:
            break
        p += 16 + _pad8(osz)
    raise Hdf5Error("global heap object %d not found in the collection at %#x" % (index, a))


def _decode(data, off, dt, esz, dims, base_addr=0):
    n = 1
    for s in (dims or ()):
        n *= s
    if dims is None:
        return None
    raw = bytes(data[off:off + n * esz])
    if isinstance(dt, tuple) and dt[0] == "V":                   # variable-length strings: (length, heap collection address, object index)
        vals = []
        for i in range(n):
            ln = int.from_bytes(raw[16 * i:16 * i + 4], "little")
            addr = int.from_bytes(raw[16 * i + 4:16 * i + 12], "little")
            idx = int.from_bytes(raw[16 * i + 12:16 * i + 16], "little")
            vals.append(b"" if (ln == 0 or addr == 0) else _global_heap_object(data, base_addr, addr, idx)[:ln])
        if not dims:
            return vals[0]
        arr = np.empty(n, dtype=object)
        arr[:] = vals
        return arr.reshape(dims)
    if isinstance(dt, tuple):                                    # fixed-length strings -> numpy 'S' array (h5py's view of it)
        arr = np.frombuffer(raw, dtype="S%d" % esz).reshape(dims)
        return arr if dims else arr.reshape(())[()]
    arr = np.frombuffer(raw, dtype=dt).reshape(dims)
    arr = arr.astype(dt.newbyteorder("="), copy=True)
    return arr if dims else arr[()]


class Dataset(_Node):
    @property
    def shape(self):
        for mtype, _f, p, _s in self._msgs:
            if mtype == 0x0001:
                return _dataspace(self.file.buf, p)
        raise Hdf5Error("dataset without a dataspace: " + self.name)

    def __array__(self, dtype=None):
        a = self[()]
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        buf = self.file.buf
        dims = dt = esz = None
        layout = None
        for mtype, _f, p, size in self._msgs:
            if mtype == 0x0001:
                dims = _dataspace(buf, p)
            elif mtype == 0x0003:
                dt, esz = _datatype(buf, p)
            elif mtype == 0x0008:
                layout = (p, size)
            elif mtype == 0x000B:
                raise NotImplementedError("filtered (compressed) dataset %s" % self.name)
        if layout is None or dt is None or dims is None:
            raise Hdf5Error("dataset %s lacks a layout / datatype / dataspace message" % self.name)
        p, _size = layout
        ver = buf.d[p]
        if ver == 3:
            lclass = buf.d[p + 1]
            if lclass == 1:                                      # contiguous
                addr, nbytes = buf.u(p + 2, 8), buf.u(p + 10, 8)
                if addr == UNDEF:
                    arr = np.zeros(dims, dtype=dt if not isinstance(dt, tuple) else "S%d" % esz)   # never written: fill value 0
                    return arr[idx] if idx != () else arr
                off = addr + self.file.base_addr
                del nbytes
            elif lclass == 0:                                    # compact: data inside the header
                off = p + 4
            else:
                raise NotImplementedError("chunked dataset %s (Keras weight files are contiguous)" % self.name)
        elif ver in (1, 2):
            rank = buf.d[p + 1]
            lclass = buf.d[p + 2]
            if lclass != 1:
                raise NotImplementedError("layout version %d class %d of dataset %s" % (ver, lclass, self.name))
            off = buf.u(p + 8, 8) + self.file.base_addr
            del rank
        else:
            raise NotImplementedError("data layout message version %d" % ver)
        arr = _decode(buf.d, off, dt, esz, dims, self.file.base_addr)
        return arr if idx == () or idx is Ellipsis else arr[idx]


class Group(_Node):
    def _links(self):
        if hasattr(self, "_cache"):
            return self._cache
        buf = self.file.buf
        links = {}
        st = [m for m in self._msgs if m[0] == 0x0011]
        if not st:
            if any(m[0] in (0x0002, 0x0006) for m in self._msgs):
                raise NotImplementedError("new-style groups (link messages; libver='latest')")
            self._cache = links
            return links
        p = st[0][2]
        btree, heap = buf.u(p, 8) + self.file.base_addr, buf.u(p + 8, 8) + self.file.base_addr
        if buf.d[heap:heap + 4] != b"HEAP":
            raise Hdf5Error("bad local heap at %#x" % heap)
        hdata = buf.u(heap + 24, 8) + self.file.base_addr

        def name_at(o):
            e = buf.d.index(b"\x00", hdata + o)
            return bytes(buf.d[hdata + o:e]).decode()

        def walk(addr):
            if buf.d[addr:addr + 4] == b"TREE":
                if buf.d[addr + 4] != 0:
                    raise Hdf5Error("B-tree node of type %d in a group" % buf.d[addr + 4])
                n = buf.u(addr + 6, 2)
                q = addr + 24                                    # sig 4, type 1, level 1, entries 2, left 8, right 8
                for i in range(n):
                    child = buf.u(q + 8 + 16 * i, 8) + self.file.base_addr       # key0, child0, key1, child1, ...
                    walk(child)
            elif buf.d[addr:addr + 4] == b"SNOD":
                n = buf.u(addr + 6, 2)
                q = addr + 8
                for i in range(n):
                    e = q + 40 * i
                    links[name_at(buf.u(e, 8))] = buf.u(e + 8, 8) + self.file.base_addr
            else:
                raise Hdf5Error("unexpected node signature %r at %#x" % (bytes(buf.d[addr:addr + 4]), addr))
        walk(btree)
        self._cache = links
        return links

    def keys(self):
        return sorted(self._links())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        node = self
        for part in [s for s in name.split("/") if s]:
            if not isinstance(node, Group):
                raise KeyError(name)
            links = node._links()
            if part not in links:
                raise KeyError(name)
            addr = links[part]
            msgs = self.file._messages(addr)
            is_group = any(m[0] == 0x0011 for m in msgs) or not any(m[0] == 0x0008 for m in msgs)
            node = (Group if is_group else Dataset)(self.file, addr, (node.name.rstrip("/") + "/" + part))
        return node


# ---- Keras weight files ---------------------------------------------------------------------------------------------
def _names(attrs, key):
    """keras saving_utils.load_attributes_from_hdf5_group: `key`, or `key0`, `key1`, ... when the list was chunked."""
    if key in attrs:
        vals = list(np.atleast_1d(attrs[key]))
    else:
        vals, i = [], 0
        while "%s%d" % (key, i) in attrs:
            vals.extend(np.atleast_1d(attrs["%s%d" % (key, i)]))
            i += 1
    return [v.decode("utf8") if isinstance(v, bytes) else str(v) for v in vals]


def load_keras_weights(path):
    """{variable name (as Keras wrote it, e.g. 'encoder/encoder_layer/dense/kernel:0'): float32 array} of a file written by
    keras `Model.save_weights(path.h5)` (hdf5_format.save_weights_to_hdf5_group: root attribute `layer_names`, one group per layer
    with attribute `weight_names` and one dataset per weight) - or of a full-model file, whose weights sit under `model_weights`.
    Also returns the layer order: (weights, [(layer name, [weight names])])."""
    f = File(path)
    root = f["model_weights"] if ("layer_names" not in f.attrs and "model_weights" in f) else f.root
    out, layout = {}, []
    for layer in _names(root.attrs, "layer_names"):
        g = root[layer]
        wn = _names(g.attrs, "weight_names")
        layout.append((layer, wn))
        for name in wn:
            out[name] = np.asarray(g[name][()], dtype=np.float32)
    return out, layout


# ---- writer: the same subset, for `Model.save_weights(path.h5)` -------------------------------------------------------------------
_LEAF_K, _NODE_K = 4, 16          # libhdf5's defaults (superblock fields): 8 symbols per SNOD, 32 children per B-tree node
_KERAS_ATTR_LIMIT = 64512         # keras HDF5_OBJECT_HEADER_LIMIT: longer name lists are split into name0, name1, ...


def _u(v, n):
    return int(v).to_bytes(n, "little")


def _dtype_message(dt):
    """Datatype message body (version 1) of a little-endian IEEE float / two's-complement integer, or ('S', n) fixed-length bytes."""
    if isinstance(dt, tuple):
        return bytes([0x13, 0x01, 0, 0]) + _u(dt[1], 4)                                     # string, null-padded, ASCII
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (2, 4, 8):
        exp_bits, man_bits = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
        bits = 8 * dt.itemsize
        return (bytes([0x11, 0x20, bits - 1, 0]) + _u(dt.itemsize, 4) + _u(0, 2) + _u(bits, 2)
                + bytes([man_bits, exp_bits, 0, man_bits]) + _u((1 << (exp_bits - 1)) - 1, 4))
    if dt.kind in "iu" and dt.itemsize in (1, 2, 4, 8):
        return bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0]) + _u(dt.itemsize, 4) + _u(0, 2) + _u(8 * dt.itemsize, 2)
    raise NotImplementedError("writing HDF5 datatype %s" % dt)


def _dataspace_message(shape):
    return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b"".join(_u(s, 8) for s in shape)      # version 1, no maximum dimensions


def _padded(b):
    return b + bytes(-len(b) % 8)


class _Writer:
    def __init__(self):
        self.buf = bytearray(96)                                   # superblock (version 0) + root symbol-table entry, filled by close()

    def put(self, data):
        self.buf.extend(bytes(-len(self.buf) % 8))
        addr = len(self.buf)
        self.buf.extend(data)
        return addr

    def header(self, messages):
        """Version-1 object header in one block: messages = [(type, flags, body)]."""
        body = b"".join(_u(t, 2) + _u(len(_padded(m)), 2) + bytes([fl, 0, 0, 0]) + _padded(m) for t, fl, m in messages)
        for _t, _fl, m in messages:
            if len(m) > 0xFFF8:
                raise Hdf5Error("object header message of %d bytes (limit 65528)" % len(m))
        return self.put(bytes([1, 0]) + _u(len(messages), 2) + _u(1, 4) + _u(len(body), 4) + bytes(4) + body)

    def attribute(self, name, value):
        """Attribute message (version 1): bytes -> fixed-length scalar string; list of bytes -> 1-D fixed-length string array;
        numpy array / scalar -> numeric."""
        nm = name.encode() + b"\x00"
        if isinstance(value, (bytes, str)):
            value = value.encode() if isinstance(value, str) else value
            dt, ds, data = _dtype_message(("S", max(len(value), 1))), _dataspace_message(()), value.ljust(max(len(value), 1), b"\x00")
        elif isinstance(value, (list, tuple)) or (isinstance(value, np.ndarray) and value.dtype.kind == "S"):
            items = [v.encode() if isinstance(v, str) else bytes(v) for v in value]
            width = max([len(v) for v in items] + [1])
            dt, ds, data = _dtype_message(("S", width)), _dataspace_message((len(items),)), b"".join(v.ljust(width, b"\x00") for v in items)
        else:
            arr = np.asarray(value, order="C")                   # (ascontiguousarray would make a scalar 1-D)
            arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
            dt, ds, data = _dtype_message(arr.dtype), _dataspace_message(arr.shape), arr.tobytes()
        return (0x000C, 0, bytes([1, 0]) + _u(len(nm), 2) + _u(len(dt), 2) + _u(len(ds), 2) + _padded(nm) + _padded(dt) + _padded(ds) + data)

    def dataset(self, array):
        arr = np.asarray(array, order="C")
        arr = arr.astype(arr.dtype.newbyteorder("<"), copy=False)
        raw = arr.tobytes()
        addr = self.put(raw) if raw else UNDEF
        return self.header([(0x0001, 0, _dataspace_message(arr.shape)), (0x0003, 1, _dtype_message(arr.dtype)),
                            (0x0005, 1, bytes([2, 2, 2, 1]) + _u(0, 4)),                  # fill value v2: late allocation, written if set, size 0
                            (0x0008, 0, bytes([3, 1]) + _u(addr, 8) + _u(len(raw), 8))])  # layout v3, contiguous

    def group(self, links, attrs=()):
        """links: {name: object header address}; attrs: [(name, value)].  -> (header address, B-tree address, heap address)."""
        names = sorted(links, key=lambda s: s.encode())
        heap_data = bytearray(8)                                   # offset 0: the empty string every group B-tree's first key points at
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data.extend(_padded(n.encode() + b"\x00"))
        data_addr = self.put(bytes(heap_data))
        heap = self.put(b"HEAP" + bytes(4) + _u(len(heap_data), 8) + _u(1, 8) + _u(data_addr, 8))    # free list: none (H5HL_FREE_NULL)
        # leaves: symbol nodes of up to 8 entries; (address, heap offset of the largest name below)
        level = []
        for i in range(0, len(names), 2 * _LEAF_K):
            part = names[i:i + 2 * _LEAF_K]
            ents = b"".join(_u(offs[n], 8) + _u(links[n], 8) + bytes(24) for n in part)
            level.append((self.put(b"SNOD" + bytes([1, 0]) + _u(len(part), 2) + ents + bytes(40 * (2 * _LEAF_K - len(part)))), offs[part[-1]]))
        depth = 0
        while True:
            nodes = []
            groups = [level[i:i + 2 * _NODE_K] for i in range(0, len(level), 2 * _NODE_K)] or [[]]
            size = 24 + 8 * (2 * _NODE_K + 1) + 8 * 2 * _NODE_K
            base = self.put(bytes(0))                               # the nodes of one level lie back to back: sibling addresses are known
            first_key = 0
            for j, kids in enumerate(groups):
                left = base + size * (j - 1) if j else UNDEF
                right = base + size * (j + 1) if j + 1 < len(groups) else UNDEF
                body = b"TREE" + bytes([0, depth]) + _u(len(kids), 2) + _u(left, 8) + _u(right, 8) + _u(first_key, 8)
                for addr, key in kids:
                    body += _u(addr, 8) + _u(key, 8)
                a = self.put(body.ljust(size, b"\x00"))
                assert a == base + size * j
                last = kids[-1][1] if kids else 0
                nodes.append((a, last))
                first_key = last
            if len(nodes) == 1:
                btree = nodes[0][0]
                break
            level, depth = nodes, depth + 1
        msgs = [(0x0011, 0, _u(btree, 8) + _u(heap, 8))] + [self.attribute(k, v) for k, v in attrs]
        return self.header(msgs), btree, heap

    def close(self, root):
        hdr, btree, heap = root
        self.buf.extend(bytes(-len(self.buf) % 8))
        sb = (_SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + _u(_LEAF_K, 2) + _u(_NODE_K, 2) + _u(0, 4)
              + _u(0, 8) + _u(UNDEF, 8) + _u(len(self.buf), 8) + _u(UNDEF, 8)
              + _u(0, 8) + _u(hdr, 8) + _u(1, 4) + _u(0, 4) + _u(btree, 8) + _u(heap, 8))
        assert len(sb) == 96
        self.buf[:96] = sb
        return bytes(self.buf)


def _name_list_attrs(key, names):
    """keras save_attributes_to_hdf5_group: one attribute, or key0, key1, ... so that each stays under the object-header limit."""
    items = [n.encode("utf8") for n in names]
    for n in items:
        if len(n) > _KERAS_ATTR_LIMIT:
            raise RuntimeError("The following attribute cannot be saved to HDF5 file because it is larger than %d bytes: %r" % (_KERAS_ATTR_LIMIT, n))
    chunks = 1
    split = [items]
    while any(len(c) * max([len(v) for v in c] + [1]) > _KERAS_ATTR_LIMIT for c in split):
        chunks += 1
        split = [list(c) for c in np.array_split(np.asarray(items, dtype=object), chunks)]
    if chunks == 1:
        return [(key, items)]
    return [("%s%d" % (key, i), c) for i, c in enumerate(split)]


def save_keras_weights(path, layers, backend="tensorflow", keras_version="2.6.0"):
    """Write what keras `Model.save_weights(path.h5)` writes (hdf5_format.save_weights_to_hdf5_group; the reference's checkpoints,
    callbacks.py:31-38,99,103): root attributes `layer_names`, `backend`, `keras_version`; one group per layer with attribute
    `weight_names`; one dataset per weight under its variable name, whose '/' make nested groups.
    layers = [(layer name, [(weight name, array)])], in model order."""
    w = _Writer()

    def build(tree, attrs):
        links = {}
        for name, node in tree.items():
            links[name] = build(node, [])[0] if isinstance(node, dict) else w.dataset(node)
        return w.group(links, attrs)

    top = {}
    for lname, weights in layers:
        if lname in top or "/" in lname:
            raise ValueError("layer name %r: duplicate, or contains '/'" % lname)
        tree = {}
        for wname, arr in weights:
            parts = [p for p in wname.split("/") if p]
            node = tree
            for p in parts[:-1]:
                node = node.setdefault(p, {})
                if not isinstance(node, dict):
                    raise ValueError("weight name %r runs through the dataset %r" % (wname, p))
            if parts[-1] in node:
                raise ValueError("duplicate weight name %r in layer %r" % (wname, lname))
            node[parts[-1]] = np.asarray(arr)
        top[lname] = (tree, _name_list_attrs("weight_names", [n for n, _ in weights]))
    links = {lname: build(tree, attrs)[0] for lname, (tree, attrs) in top.items()}
    root = w.group(links, _name_list_attrs("layer_names", [l for l, _ in layers])
                   + [("backend", backend.encode()), ("keras_version", keras_version.encode())])
    with open(path, "wb") as fh:
        fh.write(w.close(root))
