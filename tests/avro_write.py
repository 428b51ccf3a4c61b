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


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | 0x80 if n else b)
        if not n:
            return bytes(out)


def _emit_literal(out, chunk):
    ln = len(chunk) - 1
    if ln < 60:
        out.append(ln << 2)
    elif ln < 256:
        out += bytes([60 << 2, ln])
    elif ln < 65536:
        out += bytes([61 << 2, ln & 0xFF, ln >> 8])
    else:
        out += bytes([62 << 2, ln & 0xFF, (ln >> 8) & 0xFF, ln >> 16])
    out += chunk


def snappy_with_copies(data, stats=None):
    """A valid raw-snappy stream that uses ALL element kinds of the format description: literals and copies with 1-, 2- and
    4-byte offsets, overlapping copies (offset < length) included.  Greedy matcher on 4-byte hashes; the copy kind is chosen
    by what the match allows and otherwise cycled, so that every decoder branch runs (a real compressor never emits the 4-byte
    form for a near match, but it is legal and avro-rs / snappy decoders must accept it)."""
    data = bytes(data)
    out = bytearray(_varint(len(data)))
    table = {}
    i, lit_start, cyc = 0, 0, 0
    n = len(data)
    stats = stats if stats is not None else {}
    while i + 4 <= n:
        key = data[i:i + 4]
        cand = table.get(key)
        table[key] = i
        # a run of one byte: overlapping copy with offset 1
        if i > lit_start or i > 0:
            if i > 0 and data[i] == data[i - 1] and data[i:i + 4] == bytes([data[i]]) * 4:
                cand = i - 1
        if cand is None or i - cand > 0xFFFFFF:
            i += 1
            continue
        ln = 0
        while i + ln < n and data[cand + ln] == data[i + ln] and ln < 64:      # (cand + ln may run into [i, ...): overlapping copy)
            ln += 1
        if ln < 4:
            i += 1
            continue
        if i > lit_start:
            _emit_literal(out, data[lit_start:i])
        off = i - cand
        kinds = [k for k in (1, 2, 3) if (k != 1 or (4 <= ln <= 11 and off < 2048)) and (k != 2 or off < 65536)]
        kind = kinds[cyc % len(kinds)]
        cyc += 1
        if kind == 1:
            out += bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 0xFF])
        elif kind == 2:
            out += bytes([2 | ((ln - 1) << 2), off & 0xFF, off >> 8])
        else:
            out += bytes([3 | ((ln - 1) << 2), off & 0xFF, (off >> 8) & 0xFF, (off >> 16) & 0xFF, off >> 24])
        stats[kind] = stats.get(kind, 0) + 1
        if off < ln:
            stats["overlap"] = stats.get("overlap", 0) + 1
        i += ln
        lit_start = i
    if lit_start < n:
        _emit_literal(out, data[lit_start:])
    return bytes(out)


def snappy_decode(stream):
    """Independent pure-Python decoder of raw snappy (format_description.txt) -- checks the writer above without the C++ reader."""
    p, n, sh = 0, 0, 0
    while True:
        b = stream[p]; p += 1
        n |= (b & 0x7F) << sh
        sh += 7
        if not b & 0x80:
            break
    out = bytearray()
    while p < len(stream):
        tag = stream[p]; p += 1
        t = tag & 3
        if t == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(stream[p:p + nb], "little"); p += nb
            ln += 1
            out += stream[p:p + ln]; p += ln
            continue
        if t == 1:
            ln, off = 4 + ((tag >> 2) & 7), ((tag >> 5) << 8) | stream[p]; p += 1
        elif t == 2:
            ln, off = 1 + (tag >> 2), int.from_bytes(stream[p:p + 2], "little"); p += 2
        else:
            ln, off = 1 + (tag >> 2), int.from_bytes(stream[p:p + 4], "little"); p += 4
        assert 0 < off <= len(out)
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n
    return bytes(out)


def write_container(path, schema, records, codec="snappy", block_records=1000, compressor=None, corrupt=None):
    """records: already encoded bytes per record.  compressor: snappy_literal_only (default) or snappy_with_copies.
    corrupt: None | "crc" | "sync" | "truncate" -- damage the LAST block in that way (malformed-input tests)."""
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
        starts = list(range(0, len(records), block_records))
        for i in starts:
            blk = records[i:i + block_records]
            data = b"".join(blk)
            last = i == starts[-1]
            if codec == "snappy":
                crc = zlib.crc32(data) & 0xFFFFFFFF            # (of the UNCOMPRESSED bytes, big-endian: Avro spec)
                if last and corrupt == "crc":
                    crc ^= 0x5A5A5A5A
                data = (compressor or snappy_literal_only)(data) + struct.pack(">I", crc)
            whole = zz(len(blk)) + zz(len(data)) + data + (bytes(16) if last and corrupt == "sync" else sync)
            f.write(whole[:len(whole) // 2] if last and corrupt == "truncate" else whole)
