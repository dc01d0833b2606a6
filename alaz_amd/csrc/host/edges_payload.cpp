#include "edges_payload.hpp"

#include <charconv>
#include <cmath>
#include <cstdio>

namespace alaz {

// encoding/json's string encoder with escapeHTML = true (the default of json.Marshal)
void AppendJsonString(const std::string& s, std::string* out) {
    static const char kHex[] = "0123456789abcdef";
    auto u00 = [&](unsigned c) { out->append("\\u00"); out->push_back(kHex[c >> 4]); out->push_back(kHex[c & 15]); };
    out->push_back('"');
    const size_t n = s.size();
    for (size_t i = 0; i < n;) {
        const unsigned char c = (unsigned char)s[i];
        if (c < 0x80) {
            switch (c) {
            case '"': out->append("\\\""); break;
            case '\\': out->append("\\\\"); break;
            case '\n': out->append("\\n"); break;
            case '\r': out->append("\\r"); break;
            case '\t': out->append("\\t"); break;
            case '<': case '>': case '&': u00(c); break;
            default: if (c < 0x20) u00(c); else out->push_back((char)c);          // DEL (0x7f) is in encoding/json's safe set
            }
            i++; continue;
        }
        // UTF-8 validation (RFC 3629): shortest form, no surrogates, <= U+10FFFF
        size_t len = 0; uint32_t cp = 0;
        if (c >= 0xC2 && c <= 0xDF) { len = 2; cp = c & 0x1F; }
        else if (c >= 0xE0 && c <= 0xEF) { len = 3; cp = c & 0x0F; }
        else if (c >= 0xF0 && c <= 0xF4) { len = 4; cp = c & 0x07; }
        bool ok = len != 0 && i + len <= n;
        for (size_t k = 1; ok && k < len; k++) { const unsigned char d = (unsigned char)s[i + k]; if ((d & 0xC0) != 0x80) ok = false; else cp = cp << 6 | (d & 0x3F); }
        if (ok && ((len == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)))) ok = false;
        if (!ok) { out->append("\\ufffd"); i++; continue; }
        if (cp == 0x2028 || cp == 0x2029) { out->append(cp == 0x2028 ? "\\u2028" : "\\u2029"); i += len; continue; }
        out->append(s, i, len); i += len;
    }
    out->push_back('"');
}

void AppendJsonFloat(float v, std::string* out) {
    if (!std::isfinite(v)) { out->append("null"); return; }          // encoding/json refuses them; a row must not poison a payload
    char buf[48];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);          // shortest form that parses back to the same float32
    out->append(buf, r.ptr);
}

static void AppendU64(uint64_t v, std::string* out) { char buf[24]; const auto r = std::to_chars(buf, buf + sizeof buf, v); out->append(buf, r.ptr); }

std::string EdgesPayloadJson(const PayloadMetadata& md, int64_t window_end_ms, const EdgeRow* rows, size_t n) {
    std::string o;
    o.reserve(256 + n * 160);
    o += "{\"metadata\":{\"monitoring_id\":"; AppendJsonString(md.MonitoringID, &o);
    o += ",\"idempotency_key\":"; AppendJsonString(md.IdempotencyKey, &o);
    o += ",\"node_id\":"; AppendJsonString(md.NodeID, &o);
    o += ",\"alaz_version\":"; AppendJsonString(md.AlazVersion, &o);
    o += "},\"window_end\":"; { char b[24]; const auto r = std::to_chars(b, b + sizeof b, window_end_ms); o.append(b, r.ptr); }
    o += ",\"edges\":[";
    for (size_t i = 0; i < n; i++) {
        const EdgeRow& e = rows[i];
        o += i ? ",[" : "[";
        AppendJsonString(e.FromType, &o); o += ','; AppendJsonString(e.FromUID, &o); o += ',';
        AppendJsonString(e.ToType, &o); o += ','; AppendJsonString(e.ToUID, &o); o += ',';
        AppendU64(e.Count, &o); o += ','; AppendU64(e.ErrCount, &o); o += ','; AppendU64(e.SumNs, &o); o += ',';
        AppendU64(e.MaxNs, &o); o += ','; AppendU64(e.SumSqUs, &o); o += ','; AppendU64(e.Alive, &o); o += ',';
        AppendJsonFloat(e.Score, &o); o += ','; AppendJsonFloat(e.LatZ, &o); o += ','; AppendJsonFloat(e.ErrRatio, &o); o += ',';
        AppendU64(e.P50Us, &o); o += ','; AppendU64(e.P99Us, &o);          // slots 13, 14: latency percentiles in microseconds (f-3)
        o += ']';
    }
    o += "]}";
    return o;
}

int JsonEdgeSink::PersistEdges(int64_t window_end_ms, const std::vector<EdgeRow>& rows) {
    int rc = 0;                                                        // an empty window sends nothing (backend.go:606-608)
    for (size_t first = 0, k = 0; first < rows.size(); first += batch_, k++) {
        PayloadMetadata md = md_;
        md.IdempotencyKey += "-" + std::to_string(window_end_ms) + "-" + std::to_string(k);
        const size_t n = std::min(batch_, rows.size() - first);
        const int r = post_ ? post_(kEdgesEndpoint, EdgesPayloadJson(md, window_end_ms, rows.data() + first, n)) : 0;
        if (r) rc = r;
        sent_++;
    }
    return rc;
}

}  // namespace alaz
