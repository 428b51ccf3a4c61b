"""A second, independent reader of the Avro object-container format for the tests -- written from the Avro 1.11 specification's
"Object Container Files" and "Binary Encoding" sections, generic over the WRITER SCHEMA found in the file header (it knows nothing of the two index
schemas), with snappy blocks inflated by a THIRD-PARTY decompressor (pyarrow.Codec('snappy')) and the CRC-32 trailer checked with zlib.  Test-side
only.  What it is for: the files srn_avro.cpp is fed in tests/test_avro_index.py are produced by tests/avro_write.py (same author as the reader under
test); this module and pyarrow's compressor are the opinions that are not."""
import json
import struct
import zlib


class Cursor:
    def __init__(self, buf):
        self.b, self.p = memoryview(buf), 0

    def take(self, n):
        if self.p + n > len(self.b):
            raise ValueError("truncated")
        out = bytes(self.b[self.p:self.p + n])
        self.p += n
        return out

    def long(self):
        """variable-length zig-zag (spec: "int and long values are written using variable-length zig-zag coding")"""
        sh, acc = 0, 0
        while True:
            byte = self.take(1)[0]
            acc |= (byte & 0x7F) << sh
            sh += 7
            if not byte & 0x80:
                break
            if sh > 63:
                raise ValueError("varint too long")
        return (acc >> 1) ^ -(acc & 1)

    def done(self):
        return self.p == len(self.b)


def _decode(schema, c, named):
    if isinstance(schema, list):                       # union: a long index, then the value
        return _decode(schema[c.long()], c, named)
    if isinstance(schema, str):
        t = schema
        if t == "null":
            return None
        if t == "boolean":
            return c.take(1) != b"\x00"
        if t in ("int", "long"):
            return c.long()
        if t == "float":
            return struct.unpack("<f", c.take(4))[0]
        if t == "double":
            return struct.unpack("<d", c.take(8))[0]
        if t == "bytes":
            return c.take(c.long())
        if t == "string":
            return c.take(c.long()).decode()
        return _decode(named[t], c, named)
    t = schema["type"]
    if t == "record":
        named[schema["name"]] = schema
        return {f["name"]: _decode(f["type"], c, named) for f in schema["fields"]}
    if t == "array":
        out = []
        while True:
            n = c.long()
            if n == 0:
                return out
            if n < 0:                                   # a negative count is followed by the block's byte size
                n = -n
                c.long()
            out += [_decode(schema["items"], c, named) for _ in range(n)]
    if t == "map":
        out = {}
        while True:
            n = c.long()
            if n == 0:
                return out
            if n < 0:
                n = -n
                c.long()
            for _ in range(n):
                key = c.take(c.long()).decode()
                out[key] = _decode(schema["values"], c, named)
    if t == "fixed":
        return c.take(schema["size"])
    if t == "enum":
        return schema["symbols"][c.long()]
    return _decode(t, c, named)                         # {"type": "long", "logicalType": ...}


def _snappy_inflate(data):
    import pyarrow as pa
    c = Cursor(data)
    n, sh = 0, 0                                        # the raw format's preamble: uncompressed length as a plain varint
    while True:
        byte = c.take(1)[0]
        n |= (byte & 0x7F) << sh
        sh += 7
        if not byte & 0x80:
            break
    if n == 0:
        return b""
    return pa.Codec("snappy").decompress(data, decompressed_size=n).to_pybytes()


def read_container(path):
    """-> (writer schema, codec, [record dict, ...]).  Raises ValueError on anything the specification does not allow."""
    with open(path, "rb") as f:
        c = Cursor(f.read())
    if c.take(4) != b"Obj\x01":
        raise ValueError("magic")
    meta = _decode({"type": "map", "values": "bytes"}, c, {})
    sync = c.take(16)
    schema = json.loads(meta["avro.schema"])
    codec = meta.get("avro.codec", b"null").decode()
    if codec not in ("null", "snappy"):
        raise ValueError("codec " + codec)
    records = []
    while not c.done():
        count, size = c.long(), c.long()
        data = c.take(size)
        if c.take(16) != sync:
            raise ValueError("sync marker")
        if codec == "snappy":                           # "each compressed block is followed by the 4-byte, big-endian CRC32 checksum of the uncompressed data"
            body, crc = data[:-4], struct.unpack(">I", data[-4:])[0]
            data = _snappy_inflate(body)
            if zlib.crc32(data) & 0xFFFFFFFF != crc:
                raise ValueError("CRC")
        bc = Cursor(data)
        for _ in range(count):
            records.append(_decode(schema, bc, {}))
        if not bc.done():
            raise ValueError("trailing bytes in a block")
    return schema, codec, records
