// http2.cpp — see http2.hpp.  HPACK per RFC 7541 with the Write/size-update/at() behaviour of
// golang.org/x/net v0.20.0 http2/hpack; request assembly per aggregator/data.go:544-810.
#include "http2.hpp"

#include <array>
#include <cstring>
#include <memory>

namespace alaz {
namespace hpack {
namespace {

// RFC 7541 Appendix A
struct StaticEntry { const char* name; const char* value; };
const StaticEntry kStatic[61] = {
    {":authority", ""}, {":method", "GET"}, {":method", "POST"}, {":path", "/"}, {":path", "/index.html"},
    {":scheme", "http"}, {":scheme", "https"}, {":status", "200"}, {":status", "204"}, {":status", "206"},
    {":status", "304"}, {":status", "400"}, {":status", "404"}, {":status", "500"}, {"accept-charset", ""},
    {"accept-encoding", "gzip, deflate"}, {"accept-language", ""}, {"accept-ranges", ""}, {"accept", ""},
    {"access-control-allow-origin", ""}, {"age", ""}, {"allow", ""}, {"authorization", ""}, {"cache-control", ""},
    {"content-disposition", ""}, {"content-encoding", ""}, {"content-language", ""}, {"content-length", ""},
    {"content-location", ""}, {"content-range", ""}, {"content-type", ""}, {"cookie", ""}, {"date", ""}, {"etag", ""},
    {"expect", ""}, {"expires", ""}, {"from", ""}, {"host", ""}, {"if-match", ""}, {"if-modified-since", ""},
    {"if-none-match", ""}, {"if-range", ""}, {"if-unmodified-since", ""}, {"last-modified", ""}, {"link", ""},
    {"location", ""}, {"max-forwards", ""}, {"proxy-authenticate", ""}, {"proxy-authorization", ""}, {"range", ""},
    {"referer", ""}, {"refresh", ""}, {"retry-after", ""}, {"server", ""}, {"set-cookie", ""},
    {"strict-transport-security", ""}, {"transfer-encoding", ""}, {"user-agent", ""}, {"vary", ""}, {"via", ""},
    {"www-authenticate", ""}};

// RFC 7541 Appendix B: code length of symbols 0..255 and EOS (256).  The code is canonical (codes of one
// length are consecutive in symbol order, shorter codes first), so the lengths determine it.
const uint8_t kCodeLen[257] = {
    13, 23, 28, 28, 28, 28, 28, 28, 28, 24, 30, 28, 28, 30, 28, 28, 28, 28, 28, 28, 28, 28, 30, 28, 28, 28, 28, 28, 28, 28, 28, 28,
    6, 10, 10, 12, 13, 6, 8, 11, 10, 10, 8, 11, 8, 6, 6, 6, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6, 7, 8, 15, 6, 12, 10,
    13, 6, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 8, 7, 8, 13, 19, 13, 14, 6,
    15, 5, 6, 5, 6, 5, 6, 6, 6, 5, 7, 7, 6, 6, 6, 5, 6, 7, 6, 5, 5, 6, 7, 7, 7, 7, 7, 15, 11, 14, 13, 28,
    20, 22, 20, 20, 22, 22, 22, 23, 22, 23, 23, 23, 23, 23, 24, 23, 24, 24, 22, 23, 24, 23, 23, 23, 23, 21, 22, 23, 22, 23, 23, 24,
    22, 21, 20, 22, 22, 23, 23, 21, 23, 22, 22, 24, 21, 22, 23, 23, 21, 21, 22, 21, 23, 22, 23, 23, 20, 22, 22, 22, 23, 22, 22, 23,
    26, 26, 20, 19, 22, 23, 22, 25, 26, 26, 26, 27, 27, 26, 24, 25, 19, 21, 26, 27, 27, 26, 27, 24, 21, 21, 26, 26, 28, 27, 27, 27,
    20, 24, 20, 21, 22, 21, 21, 23, 22, 22, 25, 25, 24, 24, 26, 23, 26, 27, 26, 26, 27, 27, 27, 27, 27, 28, 27, 27, 27, 27, 27, 26,
    30};

// Byte-at-a-time decoding trie: a node has 256 children; a child reached by a byte whose leading bits
// complete a code is a leaf carrying the symbol and how many of the 8 bits the code used.
struct Node {
    std::unique_ptr<std::array<int32_t, 256>> child;   // index into nodes_, or -1
    uint8_t sym = 0, bits_in_last_byte = 0; bool leaf = false;
};

struct Tables {
    uint32_t code[257]; std::vector<Node> nodes;
    Tables() {
        // canonical code assignment
        uint32_t next = 0; int prev_len = 0;
        for (int len = 1; len <= 30; len++)
            for (int s = 0; s < 257; s++)
                if (kCodeLen[s] == len) { next <<= (len - prev_len); prev_len = len; code[s] = next++; }
        nodes.emplace_back(); MakeInternal(0);
        for (int s = 0; s < 256; s++) Insert((uint8_t)s, code[s], kCodeLen[s]);    // EOS is not insertable: it decodes as an error
    }
    void MakeInternal(size_t i) { nodes[i].child.reset(new std::array<int32_t, 256>()); nodes[i].child->fill(-1); }
    void Insert(uint8_t sym, uint32_t c, int len) {
        size_t cur = 0;
        while (len > 8) {
            len -= 8;
            const uint8_t b = (uint8_t)(c >> len);
            int32_t nx = (*nodes[cur].child)[b];
            if (nx < 0) { nodes.emplace_back(); nx = (int32_t)nodes.size() - 1; MakeInternal((size_t)nx); (*nodes[cur].child)[b] = nx; }
            cur = (size_t)nx;
        }
        const int shift = 8 - len;
        const uint32_t start = (c << shift) & 0xFF, count = 1u << shift;
        nodes.emplace_back(); const int32_t leaf = (int32_t)nodes.size() - 1;
        nodes[(size_t)leaf].leaf = true; nodes[(size_t)leaf].sym = sym; nodes[(size_t)leaf].bits_in_last_byte = (uint8_t)len;
        for (uint32_t k = start; k < start + count; k++) (*nodes[cur].child)[k] = leaf;
    }
};
const Tables& T() { static const Tables t; return t; }

// readVarInt
enum VarInt { kVarOk, kVarNeedMore, kVarOverflow };
VarInt ReadVarInt(unsigned n, const uint8_t* p, size_t len, uint64_t* out, size_t* used) {
    if (len == 0) return kVarNeedMore;
    const uint64_t mask = (1ull << n) - 1;
    uint64_t v = p[0] & mask;
    if (v < mask) { *out = v; *used = 1; return kVarOk; }
    unsigned shift = 0;
    for (size_t k = 1; k < len; k++) {
        v += (uint64_t)(p[k] & 0x7F) << shift;
        if (!(p[k] & 0x80)) { *out = v; *used = k + 1; return kVarOk; }
        shift += 7;
        if (shift >= 63) return kVarOverflow;
    }
    return kVarNeedMore;
}

struct RawString { const uint8_t* p = nullptr; size_t n = 0; bool huffman = false; };

// readString without a length limit
VarInt ReadString(const uint8_t* p, size_t len, RawString* s, size_t* used) {
    if (len == 0) return kVarNeedMore;
    uint64_t sl; size_t k;
    const VarInt r = ReadVarInt(7, p, len, &sl, &k);
    if (r != kVarOk) return r;
    if ((uint64_t)(len - k) < sl) return kVarNeedMore;
    s->huffman = (p[0] & 0x80) != 0; s->p = p + k; s->n = (size_t)sl; *used = k + (size_t)sl;
    return kVarOk;
}

bool DecodeString(const RawString& s, std::string* out) {
    out->clear();
    if (!s.huffman) { out->assign((const char*)s.p, s.n); return true; }
    return HuffmanDecode(s.p, s.n, out);
}

}  // namespace

bool HuffmanDecode(const uint8_t* p, size_t n, std::string* out) {
    const Tables& t = T();
    size_t node = 0;
    uint64_t acc = 0; unsigned have = 0;     // unread bits are the low `have` bits of acc
    unsigned pending = 0;                    // bits consumed into the trie since the last symbol
    for (size_t i = 0; i < n; i++) {
        acc = (acc << 8) | p[i]; have += 8;
        while (have >= 8) {
            const uint8_t b = (uint8_t)(acc >> (have - 8));
            const int32_t nx = (*t.nodes[node].child)[b];
            if (nx < 0) return false;
            const Node& c = t.nodes[(size_t)nx];
            if (c.leaf) { out->push_back((char)c.sym); have -= c.bits_in_last_byte; node = 0; pending = 0; }
            else { node = (size_t)nx; have -= 8; pending += 8; }
        }
    }
    // the last < 8 bits may still hold whole symbols
    while (have > 0) {
        const uint8_t b = (uint8_t)((acc << (8 - have)) & 0xFF);
        const int32_t nx = (*t.nodes[node].child)[b];
        if (nx < 0) return false;
        const Node& c = t.nodes[(size_t)nx];
        if (!c.leaf || c.bits_in_last_byte > have) break;
        out->push_back((char)c.sym); have -= c.bits_in_last_byte; node = 0; pending = 0;
    }
    if (pending + have > 7) return false;                           // incomplete symbol or over-long padding
    const uint64_t mask = (1ull << have) - 1;
    return (acc & mask) == mask;                                    // padding is a prefix of EOS
}

void HuffmanEncode(const std::string& s, std::string* out) {
    const Tables& t = T();
    uint64_t acc = 0; unsigned have = 0;
    for (unsigned char ch : s) {
        acc = (acc << kCodeLen[ch]) | t.code[ch]; have += kCodeLen[ch];
        while (have >= 8) { out->push_back((char)(uint8_t)(acc >> (have - 8))); have -= 8; }
        acc &= (1ull << have) - 1;
    }
    if (have) out->push_back((char)(uint8_t)((acc << (8 - have)) | ((1u << (8 - have)) - 1)));
}

bool Decoder::At(uint64_t i, std::string_view* name, std::string_view* value) const {
    if (i == 0) return false;
    if (i <= 61) { *name = kStatic[i - 1].name; *value = kStatic[i - 1].value; return true; }
    const uint64_t d = i - 62;
    if (d >= table_.size()) return false;
    *name = table_[(size_t)d].Name; *value = table_[(size_t)d].Value;
    return true;
}

void Decoder::Evict() {
    while (size_ > max_size_ && !table_.empty()) {
        size_ -= (uint32_t)(table_.back().Name.size() + table_.back().Value.size() + 32);
        table_.pop_back();
    }
}

void Decoder::Add(HeaderField f) {
    size_ += (uint32_t)(f.Name.size() + f.Value.size() + 32);
    table_.push_front(std::move(f));
    Evict();
}

Decoder::Result Decoder::ParseField(const uint8_t* p, size_t n, size_t* used) {
    const uint8_t b = p[0];
    auto from = [](VarInt v) { return v == kVarNeedMore ? kNeedMore : kError; };
    if (b & 0x80) {                                                  // 6.1 indexed header field
        uint64_t idx; size_t k;
        const VarInt r = ReadVarInt(7, p, n, &idx, &k); if (r != kVarOk) return from(r);
        std::string_view name, value;
        if (!At(idx, &name, &value)) return kError;
        *used = k;
        if (emit_) emit_(name, value);
        return kOk;
    }
    if ((b & 0xE0) == 0x20) {                                        // 6.3 dynamic table size update
        if (!first_field_ && size_ > 0) return kError;
        uint64_t sz; size_t k;
        const VarInt r = ReadVarInt(5, p, n, &sz, &k); if (r != kVarOk) return from(r);
        if (sz > allowed_max_) return kError;
        max_size_ = (uint32_t)sz; Evict();
        *used = k; return kOk;
    }
    // 6.2.x literal header field: 01xxxxxx incremental indexing, 0000xxxx without, 0001xxxx never indexed
    const bool indexing = (b & 0xC0) == 0x40;
    const unsigned prefix = indexing ? 6 : 4;
    uint64_t name_idx; size_t k;
    VarInt r = ReadVarInt(prefix, p, n, &name_idx, &k); if (r != kVarOk) return from(r);
    RawString raw_name, raw_value; size_t u;
    std::string_view name, unused;
    if (name_idx > 0) {
        if (!At(name_idx, &name, &unused)) return kError;
    } else {
        r = ReadString(p + k, n - k, &raw_name, &u); if (r != kVarOk) return from(r);
        k += u;
    }
    r = ReadString(p + k, n - k, &raw_value, &u); if (r != kVarOk) return from(r);
    k += u;
    if (name_idx == 0) { if (!DecodeString(raw_name, &name_buf_)) return kError; name = name_buf_; }
    if (!DecodeString(raw_value, &value_buf_)) return kError;
    *used = k;
    if (indexing) {
        const std::string owned(name);                               // Add() may evict the entry `name` points into
        Add(HeaderField{owned, value_buf_});
        if (emit_) emit_(owned, value_buf_);
        return kOk;
    }
    if (emit_) emit_(name, value_buf_);
    return kOk;
}

bool Decoder::Write(const uint8_t* p, size_t n) {
    if (n == 0) return true;
    std::string joined;
    if (!save_.empty()) { joined.swap(save_); joined.append((const char*)p, n); p = (const uint8_t*)joined.data(); n = joined.size(); }
    while (n > 0) {
        size_t used = 0;
        const Result r = ParseField(p, n, &used);
        if (r == kNeedMore) { save_.assign((const char*)p, n); return true; }
        first_field_ = false;
        if (r == kError) return false;
        p += used; n -= used;
    }
    return true;
}

}  // namespace hpack

uint32_t GoAtoiU32(std::string_view s) {
    size_t i = 0; bool neg = false;
    if (s.empty()) return 0;
    if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; }
    if (i == s.size()) return 0;
    const uint64_t limit = neg ? (1ull << 63) : (1ull << 63) - 1;
    uint64_t v = 0; bool clamp = false;
    for (; i < s.size(); i++) {
        const unsigned d = (unsigned)(unsigned char)s[i] - '0';
        if (d > 9) return 0;                                         // syntax error: Atoi returns 0
        if (clamp) continue;
        if (v > (limit - d) / 10) clamp = true; else v = v * 10 + d;
    }
    if (clamp) v = limit;                                            // range error: Atoi returns the nearest int64
    const int64_t r = neg ? (int64_t)(0 - v) : (int64_t)v;
    return (uint32_t)(uint64_t)r;
}

void Http2Assembler::ProcExit(uint32_t pid) {
    live_.erase(pid);
    // the reference deletes every parser whose "pid-fd" key has the decimal pid as a string prefix
    const std::string needle = std::to_string(pid);
    for (auto it = parsers_.begin(); it != parsers_.end();) {
        const std::string have = std::to_string(it->first.pid);
        if (have.compare(0, needle.size(), needle) == 0) it = parsers_.erase(it); else ++it;
    }
}

void Http2Assembler::Sweep() {
    for (auto it = frames_.begin(); it != frames_.end();) {
        if (it->second.client != it->second.server) it = frames_.erase(it); else ++it;
    }
}

bool Http2Assembler::Persist(const FrameArrival& f, const l7_req::L7Event& e, Http2Request* out) {
    if (f.method.empty() || f.path.empty()) { dropped_unparsed_++; return false; }
    const uint64_t latency = e.WriteTimeNs - f.client_write_ns;      // wraps like the Go uint64 arithmetic
    if (e.WriteTimeNs < latency) { dropped_time_++; return false; }
    out->Method = f.method; out->Path = f.path; out->Authority = f.authority;
    if (f.grpc) { out->Protocol = "gRPC"; out->StatusCode = f.grpc_status; }
    else { out->Protocol = e.Tls ? "HTTPS" : "HTTP2"; out->StatusCode = f.status; }
    out->Latency = latency;
    return true;
}

bool Http2Assembler::OnEvent(const l7_req::L7Event& e, Http2Request* out) {
    if (!live_.count(e.Pid)) { dropped_not_live_++; return false; }
    Parser& parser = parsers_[ConnKey{e.Pid, e.Fd}];                 // exists from the first event of the connection, whatever it is
    const bool from_client = e.MethodId == 1;
    if (!from_client && e.MethodId != 2) return false;
    const uint8_t* buf = e.Payload; const size_t len = e.PayloadSize;
    size_t at = 0;
    while (len - at >= 9) {
        const uint8_t* h = buf + at;
        const size_t flen = (size_t)h[0] << 16 | (size_t)h[1] << 8 | h[2];
        const bool headers = h[3] == 0x1;
        const uint32_t stream = ((uint32_t)h[5] << 24 | (uint32_t)h[6] << 16 | (uint32_t)h[7] << 8 | h[8]) & 0x7FFFFFFFu;
        at += 9;
        if (len - at < flen) break;                                  // frame cut by the 1 KiB capture
        if (!headers) { at += flen; continue; }
        const StreamKey key{e.Pid, e.Fd, stream};
        FrameArrival& f = frames_[key];
        bool complete;
        if (from_client) {
            f.client = true; f.client_write_ns = e.WriteTimeNs;
            parser.client.SetEmitFunc([&f](std::string_view name, std::string_view value) {
                if (name.empty() || (name[0] != ':' && name[0] != 'c')) return;
                if (name == ":method") { if (f.method.empty()) f.method.assign(value); }
                else if (name == ":path") { if (f.path.empty()) f.path.assign(value); }
                else if (name == ":authority") { if (f.authority.empty()) f.authority.assign(value); }
                else if (name == "content-type") { if (!f.grpc && value.substr(0, 16) == "application/grpc") f.grpc = true; }
            });
            parser.client.Write(buf + at, flen);
            parser.client.SetEmitFunc(nullptr);
            complete = f.server;
        } else {
            f.server = true;
            parser.server.SetEmitFunc([&f](std::string_view name, std::string_view value) {
                if (name == ":status") f.status = GoAtoiU32(value);
                else if (name == "grpc-status") f.grpc_status = GoAtoiU32(value);
            });
            parser.server.Write(buf + at, flen);
            parser.server.SetEmitFunc(nullptr);
            complete = f.client;
        }
        if (!complete) return false;
        const bool ok = Persist(f, e, out);
        frames_.erase(key);
        return ok;
    }
    return false;
}

}  // namespace alaz
