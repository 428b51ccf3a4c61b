"""Minimal Avro object-container WRITER for the tests (codec "null" or "snappy").  Test-side only: produces files in the
reference's pre-built index layout (src/vmisknn/vmis_index.rs:184-192, 249-255) for srn_index_new_from_avro to read."""
import json
import os
import struct
import zlib

ITEM_SCHEMA = {"type": "record", "name": "ItemIndex", "fields": [
    {"name": "ItemId", "type": "long"},
    {"name": "session_indices_time_ordered", "type": {"type": "array", "items": "int"}},
    {"name": "idf", "type": "double"},
    {"name": "ForSale", "type": "boolean"},
    {"name": "IsAdult", "type": "boolean"}]}
SESSION_SCHEMA = {"type": "record", "name": "SessionIndex", "fields": [
    {"name": "SessionIndex", "type": "int"},
    {"name": "item_ids_asc", "type": {"type": "array", "items": "long"}},
    {"name": "Time", "type": "int"}]}


def zz(n):
    n = (n << 1) ^ (n >> 63)
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def enc_array(vals):
    return (zz(len(vals)) + b"".join(zz(int(v)) for v in vals) + zz(0)) if len(vals) else zz(0)


def enc_item(item_id, sessions, idf, for_sale, adult):
    return zz(int(item_id)) + enc_array(sessions) + struct.pack("<d", float(idf)) + bytes([1 if for_sale else 0, 1 if adult else 0])


def enc_session(index, items_asc, time):
    return zz(int(index)) + enc_array(items_asc) + zz(int(time))


def snappy_literal_only(data):
    """A valid raw-snappy stream made of literals only (no back references)."""
    out = bytearray()
    n = len(data)
    while True:          # varint length
        b = n & 0x7F
        n >>= 7
        out.append(b | 0x80 if n else b)
        if not n:
            break
    for i in range(0, len(data), 65536):
        chunk = data[i:i + 65536]
        ln = len(chunk) - 1
        if ln < 60:
            out.append(ln << 2)
        elif ln < 256:
            out += bytes([60 << 2, ln])
        else:
            out += bytes([61 << 2, ln & 0xFF, ln >> 8])
        out += chunk
    return bytes(out)


def write_container(path, schema, records, codec="snappy", block_records=1000, extra_field_first=False):
    """records: already encoded bytes per record."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    sync = bytes((i * 37 + 11) & 0xFF for i in range(16))
    meta = {"avro.schema": json.dumps(schema).encode(), "avro.codec": codec.encode()}
    with open(path, "wb") as f:
        f.write(b"Obj\x01")
        f.write(zz(len(meta)))
        for k, v in meta.items():
            f.write(zz(len(k)) + k.encode() + zz(len(v)) + v)
        f.write(zz(0))
        f.write(sync)
        for i in range(0, len(records), block_records):
            blk = records[i:i + block_records]
            data = b"".join(blk)
            if codec == "snappy":
                data = snappy_literal_only(data) + struct.pack(">I", zlib.crc32(data) & 0xFFFFFFFF)
            f.write(zz(len(blk)) + zz(len(data)) + data + sync)
