// =====================================================================================
// Pre-built index in the reference's production format: two directories of Avro object-container files
// (src/vmisknn/vmis_index.rs:85-314)
//     <base>/itemindex/*.avro      {ItemId: long, session_indices_time_ordered: array<int>, idf: double, ForSale: boolean, IsAdult: boolean}   (:184-192)
//     <base>/sessionindex/*.avro   {SessionIndex: int, item_ids_asc: array<long>, Time: int}                                                     (:249-255)
// -> the flat index.  Self-contained reader: Avro container framing, codecs "null" and "snappy" (what avro-rs writes with the
// reference's Cargo features), records decoded in the WRITER schema's field order and picked by name.  Unlike the TSV path
// nothing is computed: posting lists, idf and the product flags are taken from the files (:201-228); the lists are re-ordered
// by this library's canonical recency (Time, then SessionIndex).  If every list is a most-recent prefix under that order -- what
// "time ordered, top m" lists are unless the producer broke timestamp ties differently -- the position-set kernel path applies
// to the whole index (FlatIndex::lists_complete); otherwise the lists are used as given, like the reference does, and the
// loader records PER ITEM how far its list departs from its rows (FlatIndex::viol): the prep kernel sends only the queries
// that such an item can affect to the general kernel's row pass.  Snappy blocks are checked against their CRC-32 trailer.
// =====================================================================================
#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "srn_internal.h"

namespace srn {
namespace {

struct Err { std::string msg; };
[[noreturn]] void bad(const std::string& m) { throw Err{m}; }

// ---- byte cursor with Avro's primitive encodings ---------------------------------------------
struct Cur {
    const uint8_t* p; const uint8_t* end;
    uint8_t byte() { if (p >= end) bad("truncated Avro data"); return *p++; }
    int64_t zz() {   // zig-zag varint
        uint64_t v = 0; int sh = 0;
        for (;;) { const uint8_t b = byte(); v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) break; sh += 7; if (sh > 63) bad("varint too long"); }
        return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
    }
    void need(size_t n) { if ((size_t)(end - p) < n) bad("truncated Avro data"); }
    double f64() { need(8); double d; memcpy(&d, p, 8); p += 8; return d; }
    float f32() { need(4); float d; memcpy(&d, p, 4); p += 4; return d; }
    std::string bytes() { const int64_t n = zz(); if (n < 0) bad("negative length"); need((size_t)n); std::string s((const char*)p, (size_t)n); p += n; return s; }
};

// ---- raw snappy (format_description.txt): varint uncompressed length, then literal / copy elements ----
std::vector<uint8_t> snappy_uncompress(const uint8_t* p, size_t n) {
    const uint8_t* end = p + n;
    uint64_t len = 0; int sh = 0;
    for (;;) { if (p >= end) bad("snappy: truncated"); const uint8_t b = *p++; len |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) break; sh += 7; }
    std::vector<uint8_t> out; out.reserve(len);
    while (p < end) {
        const uint8_t tag = *p++;
        if ((tag & 3) == 0) {   // literal
            size_t l = tag >> 2;
            if (l >= 60) { const int nb = (int)l - 59; if (end - p < nb) bad("snappy: truncated"); l = 0; for (int i = 0; i < nb; ++i) l |= (size_t)p[i] << (8 * i); p += nb; }
            ++l;
            if ((size_t)(end - p) < l) bad("snappy: truncated literal");
            out.insert(out.end(), p, p + l); p += l;
        } else {
            size_t l, off;
            if ((tag & 3) == 1) { if (p >= end) bad("snappy: truncated"); l = 4 + ((tag >> 2) & 7); off = ((size_t)(tag >> 5) << 8) | *p++; }
            else if ((tag & 3) == 2) { if (end - p < 2) bad("snappy: truncated"); l = 1 + (tag >> 2); off = p[0] | ((size_t)p[1] << 8); p += 2; }
            else { if (end - p < 4) bad("snappy: truncated"); l = 1 + (tag >> 2); off = p[0] | ((size_t)p[1] << 8) | ((size_t)p[2] << 16) | ((size_t)p[3] << 24); p += 4; }
            if (off == 0 || off > out.size()) bad("snappy: bad copy offset");
            const size_t from = out.size() - off;
            for (size_t i = 0; i < l; ++i) out.push_back(out[from + i]);   // (may overlap: byte by byte)
        }
    }
    if (out.size() != len) bad("snappy: length mismatch");
    return out;
}

// CRC-32 (ISO 3309 / zlib) of the uncompressed block: the 4-byte big-endian trailer of a snappy-coded Avro block
uint32_t crc32_of(const uint8_t* p, size_t n) {
    struct Table { uint32_t t[256]; Table() { for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t[i] = c; } } };
    static const Table table;   // (thread-safe initialisation)
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table.t[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// ---- the little JSON needed for a schema -----------------------------------------------------
struct Json { enum K { NUL, BOOL, NUM, STR, ARR, OBJ } k = NUL; std::string s; std::vector<Json> a; std::vector<std::pair<std::string, Json>> o;
              const Json* get(const char* key) const { for (auto& kv : o) if (kv.first == key) return &kv.second; return nullptr; } };
struct JsonParser {
    const char* p; const char* e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    Json parse() {
        ws(); if (p >= e) bad("schema: unexpected end");
        Json j;
        if (*p == '{') { j.k = Json::OBJ; ++p; ws(); if (p < e && *p == '}') { ++p; return j; }
            for (;;) { ws(); Json key = parse(); if (key.k != Json::STR) bad("schema: key"); ws(); if (p >= e || *p++ != ':') bad("schema: ':'"); j.o.emplace_back(key.s, parse()); ws();
                       if (p < e && *p == ',') { ++p; continue; } if (p < e && *p == '}') { ++p; return j; } bad("schema: object"); } }
        if (*p == '[') { j.k = Json::ARR; ++p; ws(); if (p < e && *p == ']') { ++p; return j; }
            for (;;) { j.a.push_back(parse()); ws(); if (p < e && *p == ',') { ++p; continue; } if (p < e && *p == ']') { ++p; return j; } bad("schema: array"); } }
        if (*p == '"') { j.k = Json::STR; ++p; while (p < e && *p != '"') { if (*p == '\\' && p + 1 < e) { ++p; j.s += *p++; } else j.s += *p++; } if (p >= e) bad("schema: string"); ++p; return j; }
        if (!strncmp(p, "true", 4)) { j.k = Json::BOOL; j.s = "1"; p += 4; return j; }
        if (!strncmp(p, "false", 5)) { j.k = Json::BOOL; p += 5; return j; }
        if (!strncmp(p, "null", 4)) { p += 4; return j; }
        j.k = Json::NUM; while (p < e && (isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) j.s += *p++;
        if (j.s.empty()) bad("schema: value");
        return j;
    }
};

// ---- field decoding by writer schema ---------------------------------------------------------
struct Value { int64_t i = 0; double d = 0; bool isnull = false; std::vector<int64_t> arr; };
struct Type { enum K { NUL, BOOL, INT, FLT, DBL, BYTES, ARR, UNION } k = NUL; std::vector<Type> sub; };
Type type_of(const Json& j) {
    Type t;
    if (j.k == Json::STR) {
        if (j.s == "null") t.k = Type::NUL; else if (j.s == "boolean") t.k = Type::BOOL; else if (j.s == "int" || j.s == "long") t.k = Type::INT;
        else if (j.s == "float") t.k = Type::FLT; else if (j.s == "double") t.k = Type::DBL; else if (j.s == "string" || j.s == "bytes") t.k = Type::BYTES;
        else bad("schema: unsupported type " + j.s);
    } else if (j.k == Json::ARR) { t.k = Type::UNION; for (auto& x : j.a) t.sub.push_back(type_of(x)); }
    else if (j.k == Json::OBJ) {
        const Json* ty = j.get("type"); if (!ty) bad("schema: type missing");
        if (ty->k == Json::STR && ty->s == "array") { const Json* it = j.get("items"); if (!it) bad("schema: array items"); t.k = Type::ARR; t.sub.push_back(type_of(*it)); }
        else return type_of(*ty);   // {"type": "long", "logicalType": ...}
    } else bad("schema: unsupported type");
    return t;
}
void read_value(Cur& c, const Type& t, Value& v) {
    switch (t.k) {
        case Type::NUL: v.isnull = true; break;
        case Type::BOOL: v.i = c.byte() != 0; break;
        case Type::INT: v.i = c.zz(); break;
        case Type::FLT: v.d = c.f32(); break;
        case Type::DBL: v.d = c.f64(); break;
        case Type::BYTES: c.bytes(); break;
        case Type::ARR:
            for (;;) { int64_t n = c.zz(); if (n == 0) break; if (n < 0) { n = -n; c.zz(); }   // (negative count: a byte size follows)
                       for (int64_t i = 0; i < n; ++i) { Value x; read_value(c, t.sub[0], x); v.arr.push_back(t.sub[0].k == Type::INT || t.sub[0].k == Type::BOOL ? x.i : (int64_t)x.d); } }
            break;
        case Type::UNION: { const int64_t b = c.zz(); if (b < 0 || (size_t)b >= t.sub.size()) bad("bad union branch"); read_value(c, t.sub[(size_t)b], v); break; }
    }
}

std::vector<uint8_t> slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb"); if (!f) bad("cannot open " + path);
    std::vector<uint8_t> b; uint8_t buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f); return b;
}
std::vector<std::string> avro_files(const std::string& dir) {
    std::vector<std::string> out;
    DIR* d = opendir(dir.c_str()); if (!d) bad("cannot open directory " + dir);
    while (dirent* e = readdir(d)) { const std::string n = e->d_name; if (n.size() > 5 && n.compare(n.size() - 5, 5, ".avro") == 0) out.push_back(dir + "/" + n); }
    closedir(d); std::sort(out.begin(), out.end());
    if (out.empty()) bad("no .avro files in " + dir);
    return out;
}

// every record of one container file: cb(values by the wanted field names, in the order of `want`)
template <typename F> void read_container(const std::string& path, const std::vector<std::string>& want, F cb) {
    const std::vector<uint8_t> file = slurp(path);
    Cur c{file.data(), file.data() + file.size()};
    c.need(4); if (memcmp(c.p, "Obj\x01", 4)) bad(path + ": not an Avro object container"); c.p += 4;
    std::string schema, codec = "null";
    for (;;) { int64_t n = c.zz(); if (n == 0) break; if (n < 0) { n = -n; c.zz(); }
               for (int64_t i = 0; i < n; ++i) { const std::string k = c.bytes(), v = c.bytes(); if (k == "avro.schema") schema = v; else if (k == "avro.codec") codec = v; } }
    c.need(16); uint8_t sync[16]; memcpy(sync, c.p, 16); c.p += 16;
    if (codec != "null" && codec != "snappy") bad(path + ": unsupported Avro codec " + codec);
    JsonParser jp{schema.data(), schema.data() + schema.size()}; const Json root = jp.parse();
    const Json* fields = root.get("fields"); if (!fields || fields->k != Json::ARR) bad(path + ": schema is not a record");
    std::vector<Type> types; std::vector<int> slot;   // slot[i] = index in `want` or -1
    for (const Json& f : fields->a) { const Json* nm = f.get("name"); const Json* ty = f.get("type"); if (!nm || !ty) bad(path + ": bad field");
        types.push_back(type_of(*ty)); int s = -1; for (size_t w = 0; w < want.size(); ++w) if (want[w] == nm->s) s = (int)w; slot.push_back(s); }
    for (size_t w = 0; w < want.size(); ++w) if (std::find(slot.begin(), slot.end(), (int)w) == slot.end()) bad(path + ": field " + want[w] + " missing from the schema");
    std::vector<Value> vals(want.size());
    while (c.p < c.end) {
        const int64_t count = c.zz(), size = c.zz();
        if (count < 0 || size < 0) bad(path + ": bad block header"); c.need((size_t)size);
        std::vector<uint8_t> raw; Cur b{c.p, c.p + size};
        if (codec == "snappy") {   // raw snappy + 4-byte big-endian CRC-32 of the uncompressed data
            if (size < 4) bad(path + ": short snappy block");
            raw = snappy_uncompress(c.p, (size_t)size - 4); b = Cur{raw.data(), raw.data() + raw.size()};
            const uint8_t* t = c.p + size - 4; const uint32_t want_crc = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
            if (crc32_of(raw.data(), raw.size()) != want_crc) bad(path + ": CRC mismatch in a snappy block");
        }
        for (int64_t r = 0; r < count; ++r) {
            for (auto& v : vals) v = Value();
            for (size_t f = 0; f < types.size(); ++f) { Value tmp; read_value(b, types[f], slot[f] >= 0 ? vals[(size_t)slot[f]] : tmp); }
            cb(vals);
        }
        c.p += size; c.need(16); if (memcmp(c.p, sync, 16)) bad(path + ": sync marker mismatch"); c.p += 16;
    }
}

// The recency order among sessions of EQUAL Time is not in the files; the reference compares timestamps only (vmis_index.rs:369, 404-410) and leaves ties to its
// containers.  An index built here orders ties by SessionIndex; a producer that cut its "m most recent" lists with another tie-break wrote lists that are not most-recent
// prefixes under that order -- but they are under ITS order, and any fixed order among ties is an equally valid refinement of the reference.  So the loader infers one:
// for every list, the entries that share the Time of its last entry must be more recent than the listed sessions of the same Time that hold the item and were cut.
// One virtual node per such item between the two sets, a topological order per group of equal Time, SessionIndex-descending wherever the lists say nothing (an index
// whose ties ARE by SessionIndex comes out exactly as before).  A cycle (no consistent order: ties broken per item) is broken at the largest SessionIndex and shows up
// in FlatIndex::viol.   emitted[]: the sessions of one group, most recent first.
struct TieEdge { uint32_t from, to; };
void order_tie_group(const std::vector<uint32_t>& sessions /* ascending SessionIndex */, const std::vector<uint32_t>& virt, const std::vector<uint64_t>& virt_key,
                     const std::vector<TieEdge>& edges, uint32_t ns, std::vector<uint32_t>& emitted) {
    // local ids: sessions 0..n-1 (ascending index), virtual nodes n..n+v-1
    const size_t n = sessions.size(), v = virt.size(), tot = n + v;
    auto local = [&](uint32_t g) -> size_t { return g < ns ? (size_t)(std::lower_bound(sessions.begin(), sessions.end(), g) - sessions.begin())
                                                            : n + (size_t)(std::lower_bound(virt.begin(), virt.end(), g - ns) - virt.begin()); };
    std::vector<uint32_t> indeg(tot, 0), adj_off(tot + 1, 0), adj(edges.size());
    for (const TieEdge& e : edges) { ++adj_off[local(e.from) + 1]; ++indeg[local(e.to)]; }
    for (size_t i = 0; i < tot; ++i) adj_off[i + 1] += adj_off[i];
    { std::vector<uint32_t> fill(adj_off.begin(), adj_off.end() - 1); for (const TieEdge& e : edges) adj[fill[local(e.from)]++] = (uint32_t)local(e.to); }
    auto key = [&](size_t l) -> uint64_t { return l < n ? 2ull * sessions[l] + 2ull : virt_key[l - n]; };
    std::vector<std::pair<uint64_t, uint32_t>> heap;   // max-heap of (key, local id): the ready nodes
    std::vector<uint8_t> done(tot, 0);
    for (size_t l = 0; l < tot; ++l) if (indeg[l] == 0) heap.emplace_back(key(l), (uint32_t)l);
    std::make_heap(heap.begin(), heap.end());
    size_t left = tot, scan = tot;   // scan: cycle breaker's cursor (largest key first = highest local session id first; virtual nodes never need forcing once sessions are out)
    emitted.clear();
    while (left) {
        if (heap.empty()) {   // a cycle: force the largest unfinished session
            while (scan > 0 && (done[scan - 1] || scan - 1 >= n)) --scan;
            size_t l = scan ? scan - 1 : 0; if (!scan) { for (l = n; l < tot && done[l]; ++l) {} }
            indeg[l] = 0; heap.emplace_back(key(l), (uint32_t)l); std::push_heap(heap.begin(), heap.end());
        }
        std::pop_heap(heap.begin(), heap.end()); const uint32_t l = heap.back().second; heap.pop_back();
        if (done[l]) continue;
        done[l] = 1; --left;
        if (l < n) emitted.push_back(sessions[l]);
        for (uint32_t j = adj_off[l]; j < adj_off[l + 1]; ++j) { const uint32_t t = adj[j]; if (!done[t] && indeg[t] && --indeg[t] == 0) { heap.emplace_back(key(t), t); std::push_heap(heap.begin(), heap.end()); } }
    }
}

}  // namespace

int build_flat_index_from_avro(const char* base_path, FlatIndex& ix) {
    try {
        const std::string base = base_path;
        // ---- sessions (:256-303): position = SessionIndex, unused positions are empty rows with Time 0 ----
        std::vector<std::vector<uint64_t>> rows; std::vector<uint32_t> times;
        for (const std::string& f : avro_files(base + "/sessionindex"))
            read_container(f, {"SessionIndex", "item_ids_asc", "Time"}, [&](const std::vector<Value>& v) {
                if (v[0].i < 0 || v[0].i > 0x7FFFFFF0ll) bad("SessionIndex out of range");
                const size_t s = (size_t)v[0].i;
                if (s >= rows.size()) { rows.resize(s + 1); times.resize(s + 1, 0u); }
                rows[s].assign(v[1].arr.begin(), v[1].arr.end()); times[s] = (uint32_t)v[2].i; });
        // ---- items (:193-247) ----
        struct Item { uint64_t id; std::vector<uint32_t> sessions; double idf; uint8_t attr; };
        std::vector<Item> items;
        for (const std::string& f : avro_files(base + "/itemindex"))
            read_container(f, {"ItemId", "session_indices_time_ordered", "idf", "ForSale", "IsAdult"}, [&](const std::vector<Value>& v) {
                Item it; it.id = (uint64_t)v[0].i; it.idf = v[2].d; it.attr = (uint8_t)((v[4].i ? SRN_ATTR_ADULT : 0) | (v[3].i ? SRN_ATTR_FOR_SALE : 0));
                for (int64_t s : v[1].arr) { if (s < 0 || (size_t)s >= rows.size()) bad("item " + std::to_string(it.id) + " lists a session the session index does not hold"); it.sessions.push_back((uint32_t)s); }
                items.push_back(std::move(it)); });
        const size_t ns = rows.size();
        if (ns == 0 || items.empty()) bad("empty index");
        if (items.size() >= 0xFFFFFFF0ull || ns + items.size() >= 0xFFFFFFF0ull) bad("too many items / sessions");
        ix = FlatIndex();
        // A session that no list names can never be a candidate (vmis_index.rs:332-391), so it is never a neighbour and its row is never read (mod.rs:131): like the TSV
        // builder with the sessions beyond max_session_len (:452), the loader keeps neither row nor rank for it -- a producer that writes such sessions into the
        // session index only (the CSV path keeps their rows too, :79) costs no memory, and items that occur in such rows alone need no item record
        std::vector<uint8_t> listed((ns + 7) / 8, 0);
        for (const Item& it : items) for (uint32_t s : it.sessions) listed[s >> 3] |= (uint8_t)(1u << (s & 7));
        auto is_listed = [&](size_t s) -> bool { return listed[s >> 3] >> (s & 7) & 1; };
        uint64_t n_listed = 0; for (size_t s = 0; s < ns; ++s) n_listed += is_listed(s);
        ix.n_sessions_total = ns; ix.n_kept = n_listed; ix.idf_weighting = 1.0;
        // dense idx = popularity order over the session rows (count desc, id asc), as in the TSV builder
        std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.id < b.id; });
        for (size_t i = 1; i < items.size(); ++i) if (items[i].id == items[i - 1].id) bad("item " + std::to_string(items[i].id) + " appears twice in the item index");
        auto find_item = [&](uint64_t id) -> size_t { auto it = std::lower_bound(items.begin(), items.end(), id, [](const Item& a, uint64_t v) { return a.id < v; });
                                                      return it != items.end() && it->id == id ? (size_t)(it - items.begin()) : (size_t)-1; };
        std::vector<uint32_t> cnt(items.size(), 0);
        std::vector<uint64_t> row_base(ns + 1, 0);   // rows flattened by SessionIndex: where row s starts
        std::vector<uint32_t> row_item;              // ... and its items as by-id positions (ascending, like the ids)
        for (size_t s = 0; s < ns; ++s) { uint64_t prev = 0; bool first = true;
            if (!is_listed(s)) { row_base[s + 1] = row_item.size(); std::vector<uint64_t>().swap(rows[s]); continue; }
            for (uint64_t id : rows[s]) { if (!first && id <= prev) bad("session rows must be strictly ascending item ids (item_ids_asc)"); prev = id; first = false;
                const size_t j = find_item(id); if (j == (size_t)-1) bad("item " + std::to_string(id) + " of a session row has no item-index record (the reference would panic when scoring it: vmis_index.rs:321-323)");
                ++cnt[j]; ++ix.nnz_rows; row_item.push_back((uint32_t)j); }
            row_base[s + 1] = row_item.size(); std::vector<uint64_t>().swap(rows[s]); }
        ix.total_pairs = ix.nnz_rows; ix.n_items = items.size();
        for (const Item& it : items) ix.m_index = std::max<uint64_t>(ix.m_index, it.sessions.size());
        // ---- which (session, item) pairs the lists cover; which sessions are listed at all; per item the Time of its oldest entry ----
        std::vector<uint8_t> covered((ix.nnz_rows + 7) / 8, 0), foreign(items.size(), 0);   // foreign: the list names a session whose row does not hold the item
        std::vector<uint32_t> t_old(items.size(), 0xFFFFFFFFu);
        for (size_t j = 0; j < items.size(); ++j) {
            std::vector<uint32_t> seen = items[j].sessions; std::sort(seen.begin(), seen.end());
            for (size_t e = 1; e < seen.size(); ++e) if (seen[e] == seen[e - 1]) bad("item " + std::to_string(items[j].id) + " lists a session twice");
            for (uint32_t s : items[j].sessions) {
                t_old[j] = std::min(t_old[j], times[s]);
                const uint32_t* lo = std::lower_bound(row_item.data() + row_base[s], row_item.data() + row_base[s + 1], (uint32_t)j);
                if (lo != row_item.data() + row_base[s + 1] && *lo == j) { const uint64_t b = (uint64_t)(lo - row_item.data()); covered[b >> 3] |= (uint8_t)(1u << (b & 7)); }
                else foreign[j] = 1;
            }
        }
        // ---- recency = (Time, tie order inferred from the full lists' cuts, SessionIndex where they say nothing) ----
        ix.rank_to_session.clear(); ix.rank_to_session.reserve(n_listed);
        for (size_t s = 0; s < ns; ++s) if (is_listed(s)) ix.rank_to_session.push_back((uint32_t)s);
        std::sort(ix.rank_to_session.begin(), ix.rank_to_session.end(), [&](uint32_t a, uint32_t b) { return times[a] != times[b] ? times[a] < times[b] : a < b; });
        {
            // edges of every group, as (Time, edge): kept entry -> virtual node of the item -> cut session
            struct GE { uint32_t time; TieEdge e; };
            std::vector<GE> ge; std::vector<uint8_t> has_kept(items.size(), 0), has_cut(items.size(), 0);
            std::vector<uint32_t> min_kept(items.size(), 0xFFFFFFFFu);
            for (int pass = 0; pass < 2; ++pass)   // pass 0: which items have both sides in their last entry's group; pass 1: their edges
                for (size_t s = 0; s < ns; ++s) {
                    for (uint64_t b = row_base[s]; b < row_base[s + 1]; ++b) {   // (unlisted sessions have no row here)
                        const uint32_t j = row_item[b];
                        if (times[s] != t_old[j]) continue;
                        const bool cov = covered[b >> 3] >> (b & 7) & 1;
                        if (pass == 0) { if (cov) { has_kept[j] = 1; min_kept[j] = std::min<uint32_t>(min_kept[j], (uint32_t)s); } else has_cut[j] = 1; }
                        else if (has_kept[j] && has_cut[j]) ge.push_back(GE{times[s], cov ? TieEdge{(uint32_t)s, (uint32_t)(ns + j)} : TieEdge{(uint32_t)(ns + j), (uint32_t)s}});
                    }
                }
            if (getenv("SRN_AVRO_NO_TIE_INFERENCE")) ge.clear();   // (A/B knob: ties by SessionIndex, as until round 5 -- what the per-item test then has to carry)
            std::sort(ge.begin(), ge.end(), [](const GE& a, const GE& b) { return a.time < b.time; });
            size_t g0 = 0, r0 = 0;   // r0: cursor into rank_to_session (ascending Time)
            std::vector<uint32_t> sess, virt, emitted; std::vector<uint64_t> vkey; std::vector<TieEdge> edges;
            while (g0 < ge.size()) {
                size_t g1 = g0; while (g1 < ge.size() && ge[g1].time == ge[g0].time) ++g1;
                while (times[ix.rank_to_session[r0]] != ge[g0].time) ++r0;
                size_t r1 = r0; while (r1 < n_listed && times[ix.rank_to_session[r1]] == ge[g0].time) ++r1;
                sess.assign(ix.rank_to_session.begin() + r0, ix.rank_to_session.begin() + r1);   // ascending SessionIndex
                edges.clear(); virt.clear();
                for (size_t g = g0; g < g1; ++g) { edges.push_back(ge[g].e); const TieEdge& e = ge[g].e; virt.push_back((e.from >= ns ? e.from : e.to) - (uint32_t)ns); }
                std::sort(virt.begin(), virt.end()); virt.erase(std::unique(virt.begin(), virt.end()), virt.end());
                vkey.clear(); for (uint32_t j : virt) vkey.push_back(2ull * min_kept[j] + 1ull);   // just below its lowest kept entry: an index whose ties are by SessionIndex keeps that order
                order_tie_group(sess, virt, vkey, edges, (uint32_t)ns, emitted);
                for (size_t e = 0; e < emitted.size(); ++e) ix.rank_to_session[r1 - 1 - e] = emitted[e];   // most recent first -> highest rank first
                g0 = g1; r0 = r1;
            }
        }
        std::vector<uint32_t> rank_of(ns, kNone); for (size_t r = 0; r < n_listed; ++r) rank_of[ix.rank_to_session[r]] = (uint32_t)r;
        std::vector<uint32_t> order(items.size()); std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : items[a].id < items[b].id; });
        std::vector<uint32_t> idx_of(items.size());   // by-id position -> dense idx
        ix.item_id.resize(ix.n_items); ix.id_rank.resize(ix.n_items); ix.idf.resize(ix.n_items); ix.attr.resize(ix.n_items);
        for (uint32_t i = 0; i < ix.n_items; ++i) { const uint32_t j = order[i]; idx_of[j] = i; ix.item_id[i] = items[j].id; ix.id_rank[i] = j; ix.idf[i] = items[j].idf; ix.attr[i] = items[j].attr; }
        // rows by rank
        ix.row_off.assign(n_listed + 1, 0); ix.row_items.reserve(ix.nnz_rows);
        for (size_t r = 0; r < n_listed; ++r) { const uint32_t s = ix.rank_to_session[r];
            for (uint64_t b = row_base[s]; b < row_base[s + 1]; ++b) ix.row_items.push_back(idx_of[row_item[b]]);
            ix.row_off[r + 1] = ix.row_items.size(); ix.max_row_len = std::max<uint64_t>(ix.max_row_len, row_base[s + 1] - row_base[s]); }
        ix.max_session_len = ix.max_row_len;
        // postings: the given session lists as recency ranks, most recent first; m_index = the longest list
        ix.post_off.assign(ix.n_items + 1, 0);
        for (uint32_t i = 0; i < ix.n_items; ++i) ix.post_off[i + 1] = ix.post_off[i] + items[order[i]].sessions.size();
        ix.nnz_post = ix.post_off[ix.n_items]; ix.post_rank.resize(ix.nnz_post);
        std::vector<uint32_t> oldest(ix.n_items, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < ix.n_items; ++i) {
            uint32_t* dst = &ix.post_rank[ix.post_off[i]]; const auto& ss = items[order[i]].sessions;
            for (size_t j = 0; j < ss.size(); ++j) dst[j] = rank_of[ss[j]];
            std::sort(dst, dst + ss.size(), std::greater<uint32_t>());
            if (!ss.empty()) oldest[i] = dst[ss.size() - 1];
        }
        // Which items' lists can stand in for the reference's row test?  The position-set kernels read a neighbour's first match off the lists ("r is in list_i" for
        // "row(r) contains i", mod.rs:133-138).  A session r is a VIOLATOR of item i if its row holds i, r is in no list of i, and r is LISTED (in some item's list: a
        // session no list names can never be a candidate, vmis_index.rs:332-391, so it cannot be a neighbour either).  viol[i] = 1 + the most recent violator's rank
        // (0: none; 0xFFFFFFFF: the list names a session whose row does not hold the item -- never exact).  A query is exact on the position-set path iff viol[i] <= x_lo
        // for each of its items (x_lo = the most recent m-th entry of its full lists; every neighbour is >= x_lo): DESIGN.md 4.1.  An index built here has, per item,
        // viol <= the list's last entry for a full list and 0 for a shorter one; a producer that cut its lists by a rule no tie order explains leaves larger values on
        // SOME items -- the reference uses such lists as given (vmis_index.rs:201-228), and so do we: only the queries that name such an item above their cut take the
        // general kernel's row pass.
        { std::vector<uint32_t> viol(ix.n_items, 0u);
          for (uint32_t i = 0; i < ix.n_items; ++i) if (foreign[order[i]]) viol[i] = 0xFFFFFFFFu;
          for (size_t s = 0; s < ns; ++s) {
              for (uint64_t b = row_base[s]; b < row_base[s + 1]; ++b)
                  if (!(covered[b >> 3] >> (b & 7) & 1)) { uint32_t& v = viol[idx_of[row_item[b]]]; v = std::max<uint32_t>(v, rank_of[s] + 1u); }
          }
          for (uint32_t i = 0; i < ix.n_items; ++i) { const uint64_t len = ix.post_off[i + 1] - ix.post_off[i];
              if (viol[i] != 0u && !(len == ix.m_index && viol[i] <= oldest[i])) ix.lists_complete = false; }
          if (!ix.lists_complete) ix.viol = std::move(viol); }
        size_t tcap = 16; while (tcap < ix.n_items * 2) tcap <<= 1;
        ix.id_table.assign(tcap, IdSlot{0, kNone, 0}); ix.id_mask = (uint32_t)(tcap - 1);
        for (uint32_t i = 0; i < ix.n_items; ++i) { uint32_t h = (uint32_t)mix64(ix.item_id[i]) & ix.id_mask; while (ix.id_table[h].idx != kNone) h = (h + 1) & ix.id_mask; ix.id_table[h] = IdSlot{ix.item_id[i], i, 0}; }
        return SRN_OK;
    } catch (const Err& e) { return fail(e.msg.compare(0, 11, "cannot open") == 0 ? SRN_EIO : SRN_EINVAL, std::string("avro index: ") + e.msg); }
}

}  // namespace srn
