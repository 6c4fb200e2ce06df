// flight.hip -- Arrow IPC on either side of the k-NN path, in the C ABI (host code only).
//
// f-2  lb_flight_vector_search_exchange: the body of VectorStore.handleVectorSearchExchange
//      (internal/store/vector_search_exchange.go:31-217): an Arrow IPC stream holding ONE request batch
//      {dataset utf8, k int32 (default 10), ef int32 (parsed, ignored), query_vector FixedSizeList<f32> |
//      List<f32>}, ROW 0 ONLY -> search -> an Arrow IPC stream holding {id uint64, score float32}, with the
//      reference's gRPC status codes and messages.  A Go (or any) host hands the bytes of the Flight
//      messages over and needs no Arrow glue of its own.
// f-1  lb_flight_index_add_ipc: record batches with a "vector" FixedSizeList<float32>[dim] column (and an
//      optional "id" column) appended straight from the IPC body (internal/store/store_lifecycle.go:66-76,
//      adaptive_index.go:243-256; ids truncated to the reference's uint32 VectorID, store_query.go:459-530).
//
// The IPC metadata is FlatBuffers (Message.fbs / Schema.fbs); this file carries a bounds-checked reader for
// the tables it needs and a small back-to-front builder for the two messages it writes.  Untrusted input:
// every offset is checked against the buffer before it is followed.
#include "../../include/longbow_gpu.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

// ---- gRPC status codes used by the reference handler (google.golang.org/grpc/codes) ----
enum { GRPC_OK = 0, GRPC_INVALID_ARGUMENT = 3, GRPC_NOT_FOUND = 5, GRPC_FAILED_PRECONDITION = 9, GRPC_INTERNAL = 13, GRPC_UNAVAILABLE = 14 };

struct Status {
    int code = GRPC_OK;
    std::string msg;
};
Status err(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return {code, buf};
}

// ---- bounds-checked FlatBuffers reader -----------------------------------------------------------
struct Span {
    const uint8_t *p = nullptr;
    size_t n = 0;
    bool has(size_t off, size_t len) const { return off <= n && len <= n - off; }
};
template <typename T>
bool rd(const Span &s, size_t off, T &out)
{
    if (!s.has(off, sizeof(T))) return false;
    std::memcpy(&out, s.p + off, sizeof(T));
    return true;
}

struct Table {
    Span buf;
    size_t pos = 0;      // table start
    size_t vt = 0;       // vtable start
    uint16_t vt_size = 0;
    bool ok = false;
    // byte position of field `id` inside the buffer, 0 if absent
    size_t field(int id) const
    {
        const size_t slot = 4 + 2 * (size_t)id;
        if (slot + 2 > vt_size) return 0;
        uint16_t off = 0;
        if (!rd(buf, vt + slot, off) || off == 0) return 0;
        return pos + off;
    }
    template <typename T>
    T scalar(int id, T dflt) const
    {
        const size_t f = field(id);
        T v = dflt;
        if (f) (void)rd(buf, f, v);
        return v;
    }
};
Table table_at(const Span &buf, size_t pos)
{
    Table t;
    t.buf = buf;
    int32_t so = 0;
    if (!rd(buf, pos, so)) return t;
    const int64_t vt = (int64_t)pos - so;
    if (vt < 0 || !buf.has((size_t)vt, 4)) return t;
    uint16_t vs = 0;
    if (!rd(buf, (size_t)vt, vs) || vs < 4 || !buf.has((size_t)vt, vs)) return t;
    t.pos = pos;
    t.vt = (size_t)vt;
    t.vt_size = vs;
    t.ok = true;
    return t;
}
// follow a uoffset stored at `at`
bool indirect(const Span &buf, size_t at, size_t &target)
{
    uint32_t o = 0;
    if (!at || !rd(buf, at, o)) return false;
    target = at + o;
    return target < buf.n;
}
Table subtable(const Table &t, int id)
{
    size_t tgt = 0;
    if (!indirect(t.buf, t.field(id), tgt)) return Table{};
    return table_at(t.buf, tgt);
}
// vector field: element count and the position of element 0
bool vec(const Table &t, int id, uint32_t &count, size_t &first)
{
    size_t tgt = 0;
    if (!indirect(t.buf, t.field(id), tgt)) return false;
    if (!rd(t.buf, tgt, count)) return false;
    first = tgt + 4;
    return true;
}
bool str(const Table &t, int id, std::string &out)
{
    uint32_t n = 0;
    size_t first = 0;
    if (!vec(t, id, n, first) || !t.buf.has(first, n)) return false;
    out.assign(reinterpret_cast<const char *>(t.buf.p + first), n);
    return true;
}

// ---- Arrow schema / record batch ----------------------------------------------------------------
enum TypeId { T_NONE = 0, T_NULL = 1, T_INT = 2, T_FLOAT = 3, T_BINARY = 4, T_UTF8 = 5, T_BOOL = 6, T_DECIMAL = 7, T_DATE = 8, T_TIME = 9,
              T_TIMESTAMP = 10, T_INTERVAL = 11, T_LIST = 12, T_STRUCT = 13, T_UNION = 14, T_FSBINARY = 15, T_FSLIST = 16, T_MAP = 17,
              T_DURATION = 18, T_LARGEBINARY = 19, T_LARGEUTF8 = 20, T_LARGELIST = 21 };

struct Field {
    std::string name;
    int type = T_NONE;
    int bit_width = 0;
    bool is_signed = false;
    int precision = 0; // FloatingPoint: 0 half, 1 single, 2 double
    int list_size = 0;
    std::vector<Field> children;
};

// `budget`: Field nodes this schema may still create.  FlatBuffers offsets may alias (every children vector can point
// at the same table), so per-level caps alone bound neither the work nor the memory: 64 children x depth 8 from a
// 3 KB message would be 64^8 nodes.  The total is capped instead.
constexpr int kMaxSchemaNodes = 4096;
bool parse_field(const Table &f, Field &out, int depth, int &budget)
{
    if (!f.ok || depth > 8 || --budget < 0) return false;
    (void)str(f, 0, out.name);
    out.type = f.scalar<uint8_t>(2, 0);
    const Table ty = subtable(f, 3);
    if (out.type == T_INT && ty.ok) {
        out.bit_width = ty.scalar<int32_t>(0, 0);
        out.is_signed = ty.scalar<uint8_t>(1, 0) != 0;
    } else if (out.type == T_FLOAT && ty.ok) {
        out.precision = ty.scalar<int16_t>(0, 0);
    } else if (out.type == T_FSLIST && ty.ok) {
        out.list_size = ty.scalar<int32_t>(0, 0);
    } else if (out.type == T_FSBINARY && ty.ok) {
        out.list_size = ty.scalar<int32_t>(0, 0);
    }
    uint32_t nch = 0;
    size_t first = 0;
    if (vec(f, 5, nch, first)) {
        if (nch > 64) return false;
        for (uint32_t i = 0; i < nch; i++) {
            size_t tgt = 0;
            if (!indirect(f.buf, first + 4 * (size_t)i, tgt)) return false;
            Field c;
            if (!parse_field(table_at(f.buf, tgt), c, depth + 1, budget)) return false;
            out.children.push_back(std::move(c));
        }
    }
    return true;
}

struct BufRef { int64_t off = 0, len = 0; };
struct NodeRef { int64_t len = 0, nulls = 0; };

struct Batch {
    int64_t rows = 0;
    std::vector<NodeRef> nodes;
    std::vector<BufRef> bufs;
    Span body;
};

// one decoded IPC message
struct Message {
    int header_type = 0; // 1 Schema, 2 DictionaryBatch, 3 RecordBatch
    Table header;
    Span body;
};

// Walk the stream: calls on_msg for every message until end-of-stream.  Accepts the pre-0.15 framing
// (no continuation marker) as the Arrow readers do.
template <typename F>
Status walk_stream(const uint8_t *data, size_t len, F on_msg)
{
    size_t pos = 0;
    const Span all{data, len};
    while (pos + 4 <= len) {
        uint32_t w = 0;
        (void)rd(all, pos, w);
        pos += 4;
        if (w == 0xFFFFFFFFu) {
            if (!rd(all, pos, w)) return err(GRPC_INTERNAL, "failed to read record: truncated message header");
            pos += 4;
        }
        const int32_t mlen = (int32_t)w;
        if (mlen == 0) return {}; // end of stream
        if (mlen < 0 || !all.has(pos, (size_t)mlen)) return err(GRPC_INTERNAL, "failed to read record: truncated message metadata");
        const Span meta{data + pos, (size_t)mlen};
        uint32_t root = 0;
        if (!rd(meta, 0, root)) return err(GRPC_INTERNAL, "failed to read record: bad metadata");
        const Table msg = table_at(meta, root);
        if (!msg.ok) return err(GRPC_INTERNAL, "failed to read record: bad message table");
        const int64_t body_len = msg.scalar<int64_t>(3, 0);
        pos += (size_t)mlen;
        if (body_len < 0 || !all.has(pos, (size_t)body_len)) return err(GRPC_INTERNAL, "failed to read record: truncated message body");
        Message m;
        m.header_type = msg.scalar<uint8_t>(1, 0);
        m.header = subtable(msg, 2);
        m.body = Span{data + pos, (size_t)body_len};
        pos += (size_t)body_len;
        if (!m.header.ok) return err(GRPC_INTERNAL, "failed to read record: message without header");
        bool stop = false;
        const Status st = on_msg(m, stop);
        if (st.code != GRPC_OK || stop) return st;
    }
    return {};
}

Status parse_schema(const Table &schema, std::vector<Field> &fields)
{
    if (schema.scalar<int16_t>(0, 0) != 0) return err(GRPC_INTERNAL, "failed to create record reader: big-endian stream");
    uint32_t n = 0;
    size_t first = 0;
    if (!vec(schema, 1, n, first)) return err(GRPC_INTERNAL, "failed to create record reader: schema without fields");
    if (n > (uint32_t)kMaxSchemaNodes) return err(GRPC_INTERNAL, "failed to create record reader: too many fields");
    int budget = kMaxSchemaNodes;
    for (uint32_t i = 0; i < n; i++) {
        size_t tgt = 0;
        Field f;
        if (!indirect(schema.buf, first + 4 * (size_t)i, tgt) || !parse_field(table_at(schema.buf, tgt), f, 0, budget))
            return err(GRPC_INTERNAL, "failed to create record reader: bad field %u", i);
        fields.push_back(std::move(f));
    }
    return {};
}

Status parse_batch(const Message &m, Batch &b)
{
    const Table &rb = m.header;
    b.rows = rb.scalar<int64_t>(0, 0);
    b.body = m.body;
    if (rb.field(3)) return err(GRPC_INTERNAL, "failed to read record: compressed record batches are not supported");
    uint32_t n = 0;
    size_t first = 0;
    if (vec(rb, 1, n, first)) {
        if (!rb.buf.has(first, (size_t)n * 16)) return err(GRPC_INTERNAL, "failed to read record: bad node vector");
        b.nodes.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            (void)rd(rb.buf, first + 16 * (size_t)i, b.nodes[i].len);
            (void)rd(rb.buf, first + 16 * (size_t)i + 8, b.nodes[i].nulls);
        }
    }
    if (vec(rb, 2, n, first)) {
        if (!rb.buf.has(first, (size_t)n * 16)) return err(GRPC_INTERNAL, "failed to read record: bad buffer vector");
        b.bufs.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            (void)rd(rb.buf, first + 16 * (size_t)i, b.bufs[i].off);
            (void)rd(rb.buf, first + 16 * (size_t)i + 8, b.bufs[i].len);
            if (b.bufs[i].off < 0 || b.bufs[i].len < 0 || !b.body.has((size_t)b.bufs[i].off, (size_t)b.bufs[i].len))
                return err(GRPC_INTERNAL, "failed to read record: buffer outside the message body");
        }
    }
    return {};
}

// where a column's nodes / buffers start in the flattened (depth-first) lists
struct ColPos { size_t node = 0, buf = 0; };

// advance (node, buf) past `f`; false for a layout this reader does not know
bool skip_field(const Field &f, size_t &node, size_t &buf)
{
    node += 1;
    switch (f.type) {
    case T_NULL: return true;
    case T_INT: case T_FLOAT: case T_BOOL: case T_DECIMAL: case T_DATE: case T_TIME: case T_TIMESTAMP: case T_INTERVAL:
    case T_DURATION: case T_FSBINARY:
        buf += 2;
        return true;
    case T_BINARY: case T_UTF8: case T_LARGEBINARY: case T_LARGEUTF8:
        buf += 3;
        return true;
    case T_LIST: case T_LARGELIST: case T_MAP:
        buf += 2;
        break;
    case T_FSLIST: case T_STRUCT:
        buf += 1;
        break;
    default:
        return false; // unions, views, run-end encoding ...
    }
    for (const Field &c : f.children)
        if (!skip_field(c, node, buf)) return false;
    return true;
}

bool locate(const std::vector<Field> &fields, const char *name, const Field *&f, ColPos &pos, bool &layout_ok)
{
    size_t node = 0, buf = 0;
    layout_ok = true;
    for (const Field &x : fields) {
        if (x.name == name) {
            f = &x;
            pos = {node, buf};
            return true;
        }
        if (!skip_field(x, node, buf)) {
            layout_ok = false; // a column of unknown layout precedes: positions after it are unknown
            return false;
        }
    }
    return false;
}

// buffer i of the batch if it holds at least need_bytes (>= 0) bytes, else null.  parse_batch has checked that
// [off, off + len) lies inside the message body.
const uint8_t *buf_ptr(const Batch &b, size_t i, int64_t need_bytes)
{
    if (need_bytes < 0 || i >= b.bufs.size() || b.bufs[i].len < need_bytes) return nullptr;
    return b.body.p + b.bufs[i].off;
}
// a * b with both factors >= 0, false on overflow (byte counts computed from untrusted 64-bit lengths)
bool mul_i64(int64_t a, int64_t b, int64_t &out)
{
    return a >= 0 && b >= 0 && !__builtin_mul_overflow(a, b, &out);
}
int64_t node_len(const Batch &b, size_t i) { return i < b.nodes.size() ? b.nodes[i].len : -1; }

// ---- FlatBuffers builder (back to front) -----------------------------------------------------------
struct Fbb {
    std::vector<uint8_t> buf;
    size_t head;
    size_t min_align = 1;
    size_t obj_start = 0;
    std::vector<std::pair<int, size_t>> fields; // (id, location as size() after the write)
    explicit Fbb(size_t cap = 1024) : buf(cap), head(cap) {}
    size_t size() const { return buf.size() - head; }
    void grow(size_t need)
    {
        if (head >= need) return;
        const size_t old = buf.size(), used = size();
        size_t nc = old * 2;
        while (nc - used < need) nc *= 2;
        std::vector<uint8_t> nb(nc);
        std::memcpy(nb.data() + nc - used, buf.data() + head, used);
        buf.swap(nb);
        head = nc - used;
    }
    void pad(size_t n)
    {
        grow(n);
        head -= n;
        std::memset(buf.data() + head, 0, n);
    }
    void prep(size_t align, size_t additional)
    {
        if (align > min_align) min_align = align;
        const size_t pad_n = (~(size() + additional) + 1) & (align - 1);
        pad(pad_n);
    }
    template <typename T>
    void push(T v)
    {
        grow(sizeof(T));
        head -= sizeof(T);
        std::memcpy(buf.data() + head, &v, sizeof(T));
    }
    void push_bytes(const void *p, size_t n)
    {
        grow(n);
        head -= n;
        std::memcpy(buf.data() + head, p, n);
    }
    uint32_t refer(size_t off)
    {
        prep(4, 0);
        return (uint32_t)(size() - off + 4);
    }
    size_t string(const std::string &s)
    {
        prep(4, s.size() + 1);
        push<uint8_t>(0);
        push_bytes(s.data(), s.size());
        push<uint32_t>((uint32_t)s.size());
        return size();
    }
    size_t struct_vector(const void *data, size_t elem, size_t count, size_t align)
    {
        prep(4, elem * count);
        prep(align, elem * count);
        push_bytes(data, elem * count);
        push<uint32_t>((uint32_t)count);
        return size();
    }
    size_t offset_vector(const std::vector<size_t> &offs)
    {
        prep(4, offs.size() * 4);
        for (size_t i = offs.size(); i-- > 0;) push<uint32_t>(refer(offs[i]));
        push<uint32_t>((uint32_t)offs.size());
        return size();
    }
    void start_table()
    {
        fields.clear();
        obj_start = size();
    }
    template <typename T>
    void add(int id, T v)
    {
        prep(sizeof(T), 0);
        push<T>(v);
        fields.emplace_back(id, size());
    }
    void add_offset(int id, size_t off)
    {
        const uint32_t r = refer(off);
        push<uint32_t>(r);
        fields.emplace_back(id, size());
    }
    size_t end_table()
    {
        prep(4, 0);
        push<int32_t>(0); // soffset to the vtable, patched below
        const size_t table = size();
        int nf = 0;
        for (auto &f : fields) nf = f.first + 1 > nf ? f.first + 1 : nf;
        std::vector<uint16_t> vt((size_t)nf + 2, 0);
        vt[0] = (uint16_t)((nf + 2) * 2);
        vt[1] = (uint16_t)(table - obj_start);
        for (auto &f : fields) vt[(size_t)f.first + 2] = (uint16_t)(table - f.second);
        for (size_t i = vt.size(); i-- > 0;) push<uint16_t>(vt[i]);
        const size_t vt_loc = size();
        const int32_t so = (int32_t)(vt_loc - table);
        std::memcpy(buf.data() + buf.size() - table, &so, 4);
        return table;
    }
    void finish(size_t root)
    {
        prep(min_align > 8 ? min_align : 8, 4);
        push<uint32_t>(refer(root));
    }
    const uint8_t *data() const { return buf.data() + head; }
};

void put_message(std::vector<uint8_t> &out, const Fbb &meta, const uint8_t *body, size_t body_len)
{
    const size_t mlen = meta.size();
    const size_t padded = (mlen + 7) & ~(size_t)7; // 8-byte prefix + padded metadata: the body starts 8-aligned
    const uint32_t cont = 0xFFFFFFFFu;
    const int32_t len32 = (int32_t)padded;
    const size_t at = out.size();
    out.resize(at + 8 + padded + body_len);
    std::memcpy(out.data() + at, &cont, 4);
    std::memcpy(out.data() + at + 4, &len32, 4);
    std::memcpy(out.data() + at + 8, meta.data(), mlen);
    std::memset(out.data() + at + 8 + mlen, 0, padded - mlen);
    if (body_len) std::memcpy(out.data() + at + 8 + padded, body, body_len);
}

size_t build_field(Fbb &b, const char *name, int type_id, int bit_width, bool is_signed, int precision)
{
    const size_t nm = b.string(name);
    const size_t children = b.offset_vector({}); // Arrow's reader insists on a (possibly empty) children vector
    size_t ty;
    b.start_table();
    if (type_id == T_INT) {
        b.add<int32_t>(0, bit_width);
        if (is_signed) b.add<uint8_t>(1, 1);
    } else {
        b.add<int16_t>(0, (int16_t)precision);
    }
    ty = b.end_table();
    b.start_table();
    b.add_offset(0, nm);
    b.add<uint8_t>(1, 0); // nullable = false, as array.NewRecordBuilder over the reference's schema declares it
    b.add<uint8_t>(2, (uint8_t)type_id);
    b.add_offset(3, ty);
    b.add_offset(5, children);
    return b.end_table();
}

// {id uint64, score float32} (vector_search_exchange.go:183-217) for n rows -> IPC stream bytes
void encode_results(const int64_t *ids, const float *scores, int64_t n, std::vector<uint8_t> &out)
{
    { // schema message
        Fbb b;
        const size_t f0 = build_field(b, "id", T_INT, 64, false, 0);
        const size_t f1 = build_field(b, "score", T_FLOAT, 0, false, 1);
        const size_t fv = b.offset_vector({f0, f1});
        b.start_table();
        b.add_offset(1, fv);
        const size_t schema = b.end_table();
        b.start_table();
        b.add<int16_t>(0, 4); // MetadataVersion V5
        b.add<uint8_t>(1, 1); // MessageHeader: Schema
        b.add_offset(2, schema);
        const size_t msg = b.end_table();
        b.finish(msg);
        put_message(out, b, nullptr, 0);
    }
    { // record batch message
        const size_t id_bytes = (size_t)n * 8, sc_bytes = (size_t)n * 4;
        const size_t sc_off = (id_bytes + 7) & ~(size_t)7;
        const size_t body_len = (sc_off + sc_bytes + 7) & ~(size_t)7;
        std::vector<uint8_t> body(body_len, 0);
        for (int64_t i = 0; i < n; i++) {
            const uint64_t u = (uint64_t)ids[i];
            std::memcpy(body.data() + 8 * (size_t)i, &u, 8);
        }
        if (n) std::memcpy(body.data() + sc_off, scores, sc_bytes);
        const int64_t nodes[4] = {n, 0, n, 0};
        const int64_t bufs[8] = {0, 0, 0, (int64_t)id_bytes, (int64_t)sc_off, 0, (int64_t)sc_off, (int64_t)sc_bytes};
        Fbb b;
        const size_t bv = b.struct_vector(bufs, 16, 4, 8);
        const size_t nv = b.struct_vector(nodes, 16, 2, 8);
        b.start_table();
        b.add<int64_t>(0, n);
        b.add_offset(1, nv);
        b.add_offset(2, bv);
        const size_t rb = b.end_table();
        b.start_table();
        b.add<int16_t>(0, 4);
        b.add<uint8_t>(1, 3); // MessageHeader: RecordBatch
        b.add_offset(2, rb);
        b.add<int64_t>(3, (int64_t)body_len);
        const size_t msg = b.end_table();
        b.finish(msg);
        put_message(out, b, body.data(), body_len);
    }
    const uint32_t eos[2] = {0xFFFFFFFFu, 0u};
    const size_t at = out.size();
    out.resize(at + 8);
    std::memcpy(out.data() + at, eos, 8);
}

} // namespace

struct lb_flight_datasets {
    std::mutex mu;
    std::map<std::string, lb_gpu_index *> sets;
};

namespace {
int finish(const Status &st, char *errbuf, size_t errcap)
{
    if (errbuf && errcap) {
        snprintf(errbuf, errcap, "%s", st.msg.c_str());
    }
    return st.code;
}
} // namespace

extern "C" {

lb_flight_datasets *lb_flight_datasets_new(void)
{
    try {
        return new (std::nothrow) lb_flight_datasets();
    } catch (...) {
        return nullptr;
    }
}

void lb_flight_datasets_free(lb_flight_datasets *r) { delete r; }

int lb_flight_datasets_put(lb_flight_datasets *r, const char *name, lb_gpu_index *h)
{
    if (!r || !name) return LB_ERR_INVALID_ARG;
    try {
        std::lock_guard<std::mutex> g(r->mu);
        if (h) r->sets[name] = h;
        else r->sets.erase(name);
    } catch (...) {
        return LB_ERR_OOM;
    }
    return LB_OK;
}

void lb_flight_free_buffer(uint8_t *p) { std::free(p); }

// No C++ exception may cross the C ABI (in a cgo or ctypes host that is std::terminate): every entry point below
// runs its body inside a catch-all and reports allocation failures / anything unexpected as a status.
int lb_flight_encode_results(const int64_t *ids, const float *scores, int64_t n, uint8_t **ipc_out, size_t *len_out)
{
    if (n < 0 || (n > 0 && (!ids || !scores)) || !ipc_out || !len_out) return LB_ERR_INVALID_ARG;
    try {
        while (n > 0 && ids[n - 1] < 0) n--; // fewer than k hits: the reference returns min(k, N) rows
        std::vector<uint8_t> out;
        encode_results(ids, scores, n, out);
        uint8_t *p = static_cast<uint8_t *>(std::malloc(out.size() ? out.size() : 1));
        if (!p) return LB_ERR_OOM;
        std::memcpy(p, out.data(), out.size());
        *ipc_out = p;
        *len_out = out.size();
    } catch (const std::bad_alloc &) {
        return LB_ERR_OOM;
    } catch (...) {
        return LB_ERR_INTERNAL;
    }
    return LB_OK;
}

static int exchange_body(lb_flight_datasets *reg, const uint8_t *ipc_in, size_t len_in, uint8_t **ipc_out,
                         size_t *len_out, char *errbuf, size_t errcap)
{
    if (errbuf && errcap) errbuf[0] = 0;
    if (!reg || !ipc_out || !len_out || (!ipc_in && len_in))
        return finish(err(GRPC_INVALID_ARGUMENT, "null argument"), errbuf, errcap);
    *ipc_out = nullptr;
    *len_out = 0;
    std::vector<Field> fields;
    Batch batch;
    bool have_schema = false, have_batch = false;
    Status st = walk_stream(ipc_in, len_in, [&](const Message &m, bool &stop) -> Status {
        if (m.header_type == 1) {
            if (!have_schema) {
                const Status s = parse_schema(m.header, fields);
                if (s.code != GRPC_OK) return s;
                have_schema = true;
            }
            return {};
        }
        if (m.header_type == 3) {
            if (!have_schema) return err(GRPC_INTERNAL, "failed to create record reader: record batch before schema");
            const Status s = parse_batch(m, batch);
            if (s.code != GRPC_OK) return s;
            have_batch = true;
            stop = true; // the handler reads ONE batch (reader.Next() once, vector_search_exchange.go:50)
        }
        return {};
    });
    if (st.code != GRPC_OK) return finish(st, errbuf, errcap);
    if (!have_schema) return finish(err(GRPC_INTERNAL, "failed to create record reader: no schema message"), errbuf, errcap);
    if (!have_batch) return finish(err(GRPC_INVALID_ARGUMENT, "empty search request"), errbuf, errcap);
    if (batch.rows == 0) return finish(err(GRPC_INVALID_ARGUMENT, "empty search request parameters"), errbuf, errcap);
    if (batch.rows < 0) return finish(err(GRPC_INTERNAL, "failed to read record: bad row count"), errbuf, errcap);

    const Field *f = nullptr;
    ColPos pos;
    bool layout_ok = true;
    // dataset (utf8, row 0)
    if (!locate(fields, "dataset", f, pos, layout_ok))
        return finish(layout_ok ? err(GRPC_INVALID_ARGUMENT, "missing 'dataset' column")
                                : err(GRPC_INTERNAL, "failed to read record: unsupported column type before 'dataset'"), errbuf, errcap);
    std::string name;
    if (f->type == T_UTF8 || f->type == T_BINARY) {
        const uint8_t *offs = buf_ptr(batch, pos.buf + 1, 8);
        if (!offs) return finish(err(GRPC_INTERNAL, "failed to read record: bad 'dataset' offsets"), errbuf, errcap);
        int32_t a = 0, b2 = 0;
        std::memcpy(&a, offs, 4);
        std::memcpy(&b2, offs + 4, 4);
        if (a < 0 || b2 < a) return finish(err(GRPC_INTERNAL, "failed to read record: bad 'dataset' value"), errbuf, errcap);
        const uint8_t *data = buf_ptr(batch, pos.buf + 2, b2);
        if (!data && b2 > 0) return finish(err(GRPC_INTERNAL, "failed to read record: bad 'dataset' value"), errbuf, errcap);
        if (b2 > a) name.assign(reinterpret_cast<const char *>(data + a), (size_t)(b2 - a));
    } else if (f->type == T_LARGEUTF8 || f->type == T_LARGEBINARY) {
        const uint8_t *offs = buf_ptr(batch, pos.buf + 1, 16);
        if (!offs) return finish(err(GRPC_INTERNAL, "failed to read record: bad 'dataset' offsets"), errbuf, errcap);
        int64_t a = 0, b2 = 0;
        std::memcpy(&a, offs, 8);
        std::memcpy(&b2, offs + 8, 8);
        if (a < 0 || b2 < a || b2 - a > (int64_t)1 << 20) // (a dataset name; never megabytes)
            return finish(err(GRPC_INTERNAL, "failed to read record: bad 'dataset' value"), errbuf, errcap);
        const uint8_t *data = buf_ptr(batch, pos.buf + 2, b2);
        if (!data && b2 > 0) return finish(err(GRPC_INTERNAL, "failed to read record: bad 'dataset' value"), errbuf, errcap);
        if (b2 > a) name.assign(reinterpret_cast<const char *>(data + a), (size_t)(b2 - a));
    } else {
        return finish(err(GRPC_INVALID_ARGUMENT, "'dataset' must be a string column"), errbuf, errcap);
    }
    // k (int32, default 10); ef is parsed and ignored as in the reference (:98-103,155-158)
    int k = 10;
    if (locate(fields, "k", f, pos, layout_ok)) {
        if (f->type != T_INT || (f->bit_width != 32 && f->bit_width != 64))
            return finish(err(GRPC_INVALID_ARGUMENT, "'k' must be an int32 column"), errbuf, errcap);
        const uint8_t *d = buf_ptr(batch, pos.buf + 1, f->bit_width / 8);
        if (!d) return finish(err(GRPC_INTERNAL, "failed to read record: bad 'k' buffer"), errbuf, errcap);
        if (f->bit_width == 32) { int32_t v; std::memcpy(&v, d, 4); k = v; }
        else { int64_t v; std::memcpy(&v, d, 8); k = v < 1 ? 0 : (v > (int64_t)LB_MAX_K + 1 ? LB_MAX_K + 1 : (int)v); }
    }
    // query_vector, row 0
    if (!locate(fields, "query_vector", f, pos, layout_ok))
        return finish(layout_ok ? err(GRPC_INVALID_ARGUMENT, "missing 'query_vector' column")
                                : err(GRPC_INTERNAL, "failed to read record: unsupported column type before 'query_vector'"), errbuf, errcap);
    const float *q = nullptr;
    int64_t qlen = 0;
    const bool child_f32 = f->children.size() == 1 && f->children[0].type == T_FLOAT && f->children[0].precision == 1;
    if (f->type == T_FSLIST && child_f32) {
        qlen = f->list_size;
        // buffers: [list validity][child validity][child values]; row 0 = the first list_size values
        if (qlen <= 0 || qlen > LB_MAX_DIM) return finish(err(GRPC_INTERNAL, "invalid fixed size list length"), errbuf, errcap);
        const uint8_t *vals = buf_ptr(batch, pos.buf + 2, qlen * 4);
        if (!vals) return finish(err(GRPC_INTERNAL, "invalid fixed size list length"), errbuf, errcap);
        q = reinterpret_cast<const float *>(vals);
    } else if ((f->type == T_LIST || f->type == T_LARGELIST) && child_f32) {
        // buffers: [list validity][offsets][child validity][child values]
        int64_t a = 0, b2 = 0;
        if (f->type == T_LIST) {
            const uint8_t *offs = buf_ptr(batch, pos.buf + 1, 8);
            if (!offs) return finish(err(GRPC_INTERNAL, "invalid list length"), errbuf, errcap);
            int32_t x, y;
            std::memcpy(&x, offs, 4);
            std::memcpy(&y, offs + 4, 4);
            a = x; b2 = y;
        } else {
            const uint8_t *offs = buf_ptr(batch, pos.buf + 1, 16);
            if (!offs) return finish(err(GRPC_INTERNAL, "invalid list length"), errbuf, errcap);
            std::memcpy(&a, offs, 8);
            std::memcpy(&b2, offs + 8, 8);
        }
        // 0 <= a <= b2 <= child length, where the child length is what the child's FieldNode declares AND what its
        // values buffer holds; all compared as element counts, so no byte count can overflow
        const int64_t child_vals = pos.buf + 3 < batch.bufs.size() ? batch.bufs[pos.buf + 3].len / 4 : -1;
        const int64_t child_node = node_len(batch, pos.node + 1);
        if (a < 0 || b2 < a || b2 > child_vals || (child_node >= 0 && b2 > child_node) || b2 - a > LB_MAX_DIM)
            return finish(err(GRPC_INTERNAL, "invalid list length"), errbuf, errcap);
        const uint8_t *vals = buf_ptr(batch, pos.buf + 3, b2 * 4);
        if (!vals) return finish(err(GRPC_INTERNAL, "invalid list length"), errbuf, errcap);
        q = reinterpret_cast<const float *>(vals) + a;
        qlen = b2 - a;
    } else {
        return finish(err(GRPC_INVALID_ARGUMENT, "unsupported query_vector type"), errbuf, errcap);
    }

    lb_gpu_index *h = nullptr;
    {
        std::lock_guard<std::mutex> g(reg->mu);
        auto it = reg->sets.find(name);
        if (it != reg->sets.end()) h = it->second;
    }
    if (!h) return finish(err(GRPC_NOT_FOUND, "dataset not found: %s", name.c_str()), errbuf, errcap);
    const int dim = lb_gpu_index_dim(h);
    if (qlen != dim) return finish(err(GRPC_INVALID_ARGUMENT, "dimension mismatch: expected %d, got %lld", dim, (long long)qlen), errbuf, errcap);
    if (k < 1) return finish(err(GRPC_INVALID_ARGUMENT, "k must be at least 1"), errbuf, errcap);
    // (the reference has no cap and would allocate k results; the library's search tops out at LB_MAX_K, and an
    // untrusted k must not size any buffer before it is checked)
    if (k > LB_MAX_K) return finish(err(GRPC_INVALID_ARGUMENT, "k must be at most %d", LB_MAX_K), errbuf, errcap);
    std::vector<float> qa((size_t)qlen); // (the values buffer is only as aligned as the caller's bytes)
    std::memcpy(qa.data(), reinterpret_cast<const void *>(q), (size_t)qlen * sizeof(float));
    std::vector<float> dist((size_t)k);
    std::vector<int64_t> labels((size_t)k);
    const int rc = lb_gpu_index_search(h, 1, qa.data(), k, dist.data(), labels.data());
    if (rc == LB_ERR_NO_DEVICE) return finish(err(GRPC_UNAVAILABLE, "search failed: GPU not available"), errbuf, errcap);
    if (rc != LB_OK) return finish(err(GRPC_INTERNAL, "search failed: %s (%s)", lb_gpu_status_string(rc), lb_gpu_last_error(h)), errbuf, errcap);
    const int erc = lb_flight_encode_results(labels.data(), dist.data(), k, ipc_out, len_out);
    if (erc != LB_OK) return finish(err(GRPC_INTERNAL, "failed to write response"), errbuf, errcap);
    return GRPC_OK;
}

int lb_flight_vector_search_exchange(lb_flight_datasets *reg, const uint8_t *ipc_in, size_t len_in, uint8_t **ipc_out,
                                     size_t *len_out, char *errbuf, size_t errcap)
{
    try {
        return exchange_body(reg, ipc_in, len_in, ipc_out, len_out, errbuf, errcap);
    } catch (const std::bad_alloc &) {
        return finish(err(GRPC_INTERNAL, "out of memory while handling the request"), errbuf, errcap);
    } catch (...) {
        return finish(err(GRPC_INTERNAL, "internal error while handling the request"), errbuf, errcap);
    }
}

static int add_ipc_body(lb_gpu_index *h, const uint8_t *ipc, size_t len, int64_t *rows_added, char *errbuf, size_t errcap)
{
    if (errbuf && errcap) errbuf[0] = 0;
    if (rows_added) *rows_added = 0;
    if (!h || (!ipc && len)) return finish(err(GRPC_INVALID_ARGUMENT, "null argument"), errbuf, errcap);
    const int dim = lb_gpu_index_dim(h);
    std::vector<Field> fields;
    bool have_schema = false;
    int64_t added = 0;
    bool use_ids = false; // once a batch carried ids, later batches without an id column report positions
    Status st = walk_stream(ipc, len, [&](const Message &m, bool &) -> Status {
        if (m.header_type == 1) {
            if (!have_schema) {
                const Status s = parse_schema(m.header, fields);
                if (s.code != GRPC_OK) return s;
                have_schema = true;
            }
            return {};
        }
        if (m.header_type != 3) return {};
        if (!have_schema) return err(GRPC_INTERNAL, "record batch before schema");
        Batch b;
        const Status s = parse_batch(m, b);
        if (s.code != GRPC_OK) return s;
        if (b.rows == 0) return {};
        if (b.rows < 0 || b.rows > (int64_t)0xffffffffll) return err(GRPC_INTERNAL, "failed to read record: bad row count");
        const Field *f = nullptr;
        ColPos pos;
        bool layout_ok = true;
        if (!locate(fields, "vector", f, pos, layout_ok))
            return layout_ok ? err(GRPC_INVALID_ARGUMENT, "missing 'vector' column") : err(GRPC_INTERNAL, "unsupported column type before 'vector'");
        if (f->type != T_FSLIST || f->children.size() != 1)
            return err(GRPC_INVALID_ARGUMENT, "'vector' must be FixedSizeList");
        if (f->list_size != dim) return err(GRPC_INVALID_ARGUMENT, "dimension mismatch: expected %d, got %d", dim, f->list_size);
        if (f->children[0].type != T_FLOAT || f->children[0].precision != 1)
            return err(GRPC_INVALID_ARGUMENT, "'vector' elements must be float32 in this entry point");
        if (pos.node < b.nodes.size() && b.nodes[pos.node].nulls != 0) return err(GRPC_INVALID_ARGUMENT, "null vectors are not supported");
        // the column's FieldNodes must agree with the batch: `rows` lists, rows * dim child values
        int64_t nvals = 0, vbytes = 0;
        if (!mul_i64(b.rows, dim, nvals) || !mul_i64(nvals, 4, vbytes)) return err(GRPC_INTERNAL, "'vector' values buffer is too short");
        if ((node_len(b, pos.node) >= 0 && node_len(b, pos.node) != b.rows) ||
            (node_len(b, pos.node + 1) >= 0 && node_len(b, pos.node + 1) != nvals))
            return err(GRPC_INTERNAL, "failed to read record: 'vector' lengths disagree with the batch");
        const uint8_t *vals = buf_ptr(b, pos.buf + 2, vbytes);
        if (!vals) return err(GRPC_INTERNAL, "'vector' values buffer is too short");
        std::vector<int64_t> ids;
        const Field *fi = nullptr;
        ColPos pi;
        if (locate(fields, "id", fi, pi, layout_ok) && fi->type == T_INT && (fi->bit_width == 64 || (fi->bit_width == 32 && !fi->is_signed))) {
            const int w = fi->bit_width / 8;
            int64_t idbytes = 0;
            const uint8_t *d = mul_i64(b.rows, w, idbytes) ? buf_ptr(b, pi.buf + 1, idbytes) : nullptr;
            if (!d) return err(GRPC_INTERNAL, "'id' buffer is too short");
            ids.resize((size_t)b.rows);
            for (int64_t i = 0; i < b.rows; i++) {
                uint64_t v = 0;
                std::memcpy(&v, d + (size_t)i * w, (size_t)w);
                ids[(size_t)i] = (int64_t)(v & 0xFFFFFFFFull); // core.VectorID is uint32 (store_query.go:505-530)
            }
            use_ids = true;
        } else if (use_ids) {
            ids.resize((size_t)b.rows);
            const int64_t base = lb_gpu_index_ntotal(h);
            for (int64_t i = 0; i < b.rows; i++) ids[(size_t)i] = base + i;
        }
        // the values buffer goes to the library as it lies in the IPC body: no repacking
        const int rc = lb_gpu_index_add(h, b.rows, reinterpret_cast<const float *>(vals), ids.empty() ? nullptr : ids.data());
        if (rc != LB_OK) return err(rc == LB_ERR_NO_DEVICE ? GRPC_UNAVAILABLE : GRPC_INTERNAL, "add failed: %s (%s)", lb_gpu_status_string(rc), lb_gpu_last_error(h));
        added += b.rows;
        return {};
    });
    if (rows_added) *rows_added = added;
    if (st.code == GRPC_OK && !have_schema) st = err(GRPC_INTERNAL, "no schema message");
    return finish(st, errbuf, errcap);
}

int lb_flight_index_add_ipc(lb_gpu_index *h, const uint8_t *ipc, size_t len, int64_t *rows_added, char *errbuf, size_t errcap)
{
    try {
        return add_ipc_body(h, ipc, len, rows_added, errbuf, errcap);
    } catch (const std::bad_alloc &) {
        return finish(err(GRPC_INTERNAL, "out of memory while reading the stream"), errbuf, errcap);
    } catch (...) {
        return finish(err(GRPC_INTERNAL, "internal error while reading the stream"), errbuf, errcap);
    }
}

} // extern "C"
