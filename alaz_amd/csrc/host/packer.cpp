#include "packer.hpp"

#include <cstdio>
#include <cstring>

namespace alaz {

void ParseHttpPayload(const char* req, size_t len, std::string* method, std::string* path, std::string* version, std::string* host) {
    method->clear(); path->clear(); version->clear(); host->clear();
    size_t l0 = 0;
    while (l0 < len && req[l0] != '\n') l0++;
    // parts := strings.Split(lines[0], " "); if len(parts) >= 3 { method, path, httpVersion = parts[0..2] }
    size_t st[3] = {0, 0, 0}, en[3] = {0, 0, 0}, np = 0, s = 0;
    for (size_t i = 0; i <= l0; i++)
        if (i == l0 || req[i] == ' ') { if (np < 3) { st[np] = s; en[np] = i; } np++; s = i + 1; }
    if (np >= 3) { method->assign(req + st[0], en[0] - st[0]); path->assign(req + st[1], en[1] - st[1]); version->assign(req + st[2], en[2] - st[2]); }
    if (l0 >= len) return;                       // single line: lines[1:] is empty
    size_t pos = l0 + 1;
    for (;;) {
        size_t e = pos;
        while (e < len && req[e] != '\n') e++;
        const size_t n = e - pos;
        if (n >= 5 && std::memcmp(req + pos, "Host:", 5) == 0) {
            const char* sp = static_cast<const char*>(std::memchr(req + pos, ' ', n));
            if (sp) {                                                   // len(hostParts) >= 2
                const char* b = sp + 1; const char* lim = req + e; const char* q = b;
                while (q < lim && *q != ' ') q++;
                size_t hn = (size_t)(q - b);
                if (hn > 0 && b[hn - 1] == '\r') hn--;                  // strings.TrimSuffix(hostHeader, "\r")
                host->assign(b, hn);
                return;                                                 // break
            }
        }
        if (e >= len) return;
        pos = e + 1;
    }
}

bool ContainsSQLKeywords(const uint8_t* s, size_t n) {
    static const char* kw[] = {"SELECT", "INSERT INTO", "UPDATE", "DELETE FROM", "CREATE TABLE", "ALTER TABLE", "DROP TABLE", "TRUNCATE TABLE",
                               "BEGIN", "COMMIT", "ROLLBACK", "SAVEPOINT", "CREATE INDEX", "DROP INDEX", "CREATE VIEW", "DROP VIEW", "GRANT", "REVOKE", "EXECUTE"};
    std::string up; up.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const uint8_t c = s[i];
        if (c == 0xC4 && i + 1 < n && s[i + 1] == 0xB1) { up.push_back('I'); i++; }          // strings.ToUpper: U+0131 -> 'I'
        else if (c == 0xC5 && i + 1 < n && s[i + 1] == 0xBF) { up.push_back('S'); i++; }     // U+017F -> 'S'
        else up.push_back((c >= 'a' && c <= 'z') ? (char)(c - 32) : (char)c);
    }
    for (const char* k : kw) if (up.find(k) != std::string::npos) return true;
    return false;
}

uint32_t L7Packer::InternLabel(const std::string& host) {
    auto it = label_ids_.find(host);
    if (it != label_ids_.end()) return it->second;
    labels_.push_back(host);
    const uint32_t id = (uint32_t)labels_.size();       // ids start at 1; 0 = no label
    label_ids_.emplace(host, id);
    return id;
}

static std::vector<std::pair<size_t, size_t>> SplitNul(const uint8_t* b, size_t n) {    // bytes.Split(b, []byte{0})
    std::vector<std::pair<size_t, size_t>> v; size_t s = 0;
    for (size_t i = 0; i <= n; i++) if (i == n || b[i] == 0) { v.emplace_back(s, i - s); s = i + 1; }
    return v;
}

// parsePostgresCommand — aggregator/data.go:1474-1556.  0 = ok, -1 = error (event dropped :1328-1332)
int L7Packer::ParsePostgres(const l7_req::L7Event& e, std::string* out) {
    const uint8_t* r = e.Payload; size_t n = e.PayloadSize;
    const std::string method = e.Method();
    out->clear();
    if (method == "SIMPLE_QUERY") {
        if (n < 5) return -1;
        r += 5; n -= 5;
        if (!ContainsSQLKeywords(r, n)) return -1;
        out->assign((const char*)r, n);
        return 0;
    }
    if (method == "EXTENDED_QUERY") {
        if (n < 5) return -1;
        const uint8_t id = r[0];
        auto vars = SplitNul(r + 5, n - 5);
        char key[64];
        std::snprintf(key, sizeof key, "%u-%llu-", e.Pid, (unsigned long long)e.Fd);
        if (id == 'P') {
            std::string name, query;
            if (vars.size() >= 3) { name.assign((const char*)r + 5 + vars[0].first, vars[0].second); query.assign((const char*)r + 5 + vars[1].first, vars[1].second); }
            else if (vars.size() == 2) { name.assign((const char*)r + 5 + vars[0].first, vars[0].second); query.assign((const char*)r + 5 + vars[1].first, vars[1].second); query += "..."; }
            else return -1;
            pg_stmts_[std::string(key) + name] = query;
            *out = "PREPARE " + name + " AS " + query;
            return 0;
        }
        if (id == 'B') {
            if (vars.size() < 2) return -1;
            const std::string name((const char*)r + 5 + vars[1].first, vars[1].second);
            auto it = pg_stmts_.find(std::string(key) + name);
            if (it == pg_stmts_.end() || it->second.empty()) { *out = "EXECUTE " + name + " *values*"; return 0; }
            *out = it->second;
            return 0;
        }
        return -1;
    }
    if (method == "CLOSE_OR_TERMINATE") out->assign((const char*)r, n);
    return 0;
}

// parseMySQLCommand — aggregator/data.go:1431-1472
int L7Packer::ParseMySQL(const l7_req::L7Event& e, std::string* out) {
    const uint8_t* r = e.Payload; size_t n = e.PayloadSize;
    out->clear();
    if (n < 5) return -1;
    r += 5; n -= 5;
    const std::string method = e.Method();
    char key[96];
    if (method == "TEXT_QUERY") { if (!ContainsSQLKeywords(r, n)) return -1; }
    else if (method == "PREPARE_STMT") {
        std::snprintf(key, sizeof key, "%u-%llu-%u", e.Pid, (unsigned long long)e.Fd, e.MySqlPrepStmtId);
        mysql_stmts_[key] = std::string((const char*)r, n);
    } else if (method == "EXEC_STMT" || method == "STMT_CLOSE") {
        if (n < 4) return 0;
        const uint32_t sid = (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
        std::snprintf(key, sizeof key, "%u-%llu-%u", e.Pid, (unsigned long long)e.Fd, sid);
        if (method[0] == 'E') {
            auto it = mysql_stmts_.find(key);
            if (it == mysql_stmts_.end() || it->second.empty()) { *out = "EXECUTE " + std::to_string(sid) + " *values*"; return 0; }
            *out = it->second; return 0;
        }
        mysql_stmts_.erase(key);
        *out = "CLOSE STMT " + std::to_string(sid) + " ";
        return 0;
    }
    out->assign((const char*)r, n);
    return 0;
}

// parseMongoEvent — aggregator/data.go:1561-1617: out-of-range slices panic, the deferred recover()
// (:1562-1567) turns that into ("", nil) => the event is persisted with an empty path.
int L7Packer::ParseMongo(const l7_req::L7Event& e, std::string* out) {
    const uint8_t* p = e.Payload; size_t n = e.PayloadSize;
    out->clear();
#define SG_PANIC_IF(c) do { if (c) return 0; } while (0)
    SG_PANIC_IF(n < 12); p += 12; n -= 12;
    SG_PANIC_IF(n < 4);
    const uint32_t opcode = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    SG_PANIC_IF(n < 8); p += 8; n -= 8;
    if (opcode == 2012) { *out = "compressed mongo event"; return 0; }
    if (opcode == 2013) {
        SG_PANIC_IF(n < 1);
        const uint8_t kind = p[0]; p += 1; n -= 1;
        if (kind == 0) {
            SG_PANIC_IF(n < 4);
            const uint32_t doc_len = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
            SG_PANIC_IF(doc_len < 4 || doc_len > n);
            p += 4; n = doc_len - 4;
            SG_PANIC_IF(n < 1);
            if (p[0] != 2) return -1;
            p += 1; n -= 1;
            size_t el = 0; while (el < n && p[el] != 0) el++;
            SG_PANIC_IF(el + 5 > n);
            const uint32_t vlen = (uint32_t)p[el + 1] | (uint32_t)p[el + 2] << 8 | (uint32_t)p[el + 3] << 16 | (uint32_t)p[el + 4] << 24;
            SG_PANIC_IF(vlen == 0 || vlen - 1 > n - (el + 5));
            *out = std::string((const char*)p, el) + " " + std::string((const char*)p + el + 5, vlen - 1);
            return 0;
        }
    }
#undef SG_PANIC_IF
    return -1;
}

void L7Packer::ConnClosed(uint32_t pid, uint64_t fd) {
    h2_.ConnClosed(pid, fd);
    const std::string prefix = std::to_string(pid) + "-" + std::to_string(fd);
    for (auto it = pg_stmts_.begin(); it != pg_stmts_.end();) {
        if (it->first.compare(0, prefix.size(), prefix) == 0) it = pg_stmts_.erase(it); else ++it;
    }
}

void L7Packer::ProcExit(uint32_t pid) {
    h2_.ProcExit(pid);
    const std::string prefix = std::to_string(pid);
    for (auto it = pg_stmts_.begin(); it != pg_stmts_.end();) {
        if (it->first.compare(0, prefix.size(), prefix) == 0) it = pg_stmts_.erase(it); else ++it;
    }
}

size_t L7Packer::PackWire(const uint8_t* rec, uint32_t kafka_msgs, std::vector<sg_event>* out) {
    using namespace l7_req;
    L7Event& e = scratch_;
    DecodeWire(rec, &e, /*copy_payload=*/false);
    bool needs_payload;
    switch (e.ProtocolId) {
    case BPF_L7_PROTOCOL_HTTP: needs_payload = !IsKnownIP(e.Daddr); break;
    case BPF_L7_PROTOCOL_POSTGRES: case BPF_L7_PROTOCOL_MYSQL: case BPF_L7_PROTOCOL_MONGO: case BPF_L7_PROTOCOL_HTTP2: needs_payload = true; break;
    case BPF_L7_PROTOCOL_KAFKA: needs_payload = kafka_decode_; break;
    default: needs_payload = false;
    }
    if (needs_payload) {
        std::memcpy(e.Payload, rec + 36, e.PayloadSize);
        if (e.PayloadSize < kMaxPayload) e.Payload[e.PayloadSize] = 0;          // handlers only read [0, PayloadSize)
    }
    return Pack(e, kafka_msgs, out);
}

size_t L7Packer::Pack(const l7_req::L7Event& e, uint32_t kafka_msgs, std::vector<sg_event>* out) {
    using namespace l7_req;
    sg_event ev;
    ev.saddr = e.Saddr; ev.daddr = e.Daddr; ev.host_label = 0;
    ev.status = (uint16_t)(e.Status > 0xFFFFu ? 0xFFFFu : e.Status);
    ev.protocol = e.ProtocolId; ev.flags = e.Tls ? SG_EV_TLS : 0;
    ev.duration_ns = e.Duration; ev.write_time_ns = e.WriteTimeNs;
    std::string tmp;
    switch (e.ProtocolId) {
    case BPF_L7_PROTOCOL_HTTP: {
        // the Host header is only ever used when daddr is neither a service nor a pod IP
        if (!IsKnownIP(e.Daddr)) {
            std::string m, p, v, host;
            ParseHttpPayload((const char*)e.Payload, e.PayloadSize, &m, &p, &v, &host);
            if (!host.empty()) ev.host_label = InternLabel(host);
        }
        out->push_back(ev); return 1;
    }
    case BPF_L7_PROTOCOL_POSTGRES:
        if (ParsePostgres(e, &tmp) != 0) { dropped_parse_++; return 0; }
        out->push_back(ev); return 1;
    case BPF_L7_PROTOCOL_MYSQL:
        if (ParseMySQL(e, &tmp) != 0) { dropped_parse_++; return 0; }
        out->push_back(ev); return 1;
    case BPF_L7_PROTOCOL_MONGO:
        if (ParseMongo(e, &tmp) != 0) { dropped_parse_++; return 0; }
        out->push_back(ev); return 1;
    case BPF_L7_PROTOCOL_REDIS:
        if (e.MethodId == 2) ev.flags |= SG_EV_REVERSE;                 // PUSHED_EVENT
        out->push_back(ev); return 1;
    case BPF_L7_PROTOCOL_AMQP:
        if (e.MethodId == 2) ev.flags |= SG_EV_REVERSE;                 // DELIVER
        out->push_back(ev); return 1;
    case BPF_L7_PROTOCOL_KAFKA:
        if (e.MethodId == 2) ev.flags |= SG_EV_CONSUME;
        if (kafka_decode_) {                                            // decodeKafkaPayload, data.go:929-1017
            size_t n = 0;
            if (kafka::DecodePayload(e.Payload, e.PayloadSize, e.MethodId, e.KafkaApiVersion, nullptr, &n) != kafka::Status::kOk) n = 0;
            if (n == 0) { dropped_parse_++; return 0; }
            kafka_msgs = (uint32_t)n;
        }
        for (uint32_t k = 0; k < kafka_msgs; k++) out->push_back(ev);    // one KafkaEvent per decoded message
        return kafka_msgs;
    case BPF_L7_PROTOCOL_HTTP2: {
        Http2Request r;
        if (!h2_.OnEvent(e, &r)) return 0;                              // not (yet) a complete request, or dropped
        ev.duration_ns = r.Latency;
        ev.status = (uint16_t)(r.StatusCode > 0xFFFFu ? 0xFFFFu : r.StatusCode);
        if (!IsKnownIP(e.Daddr) && !r.Authority.empty()) ev.host_label = InternLabel(r.Authority);
        out->push_back(ev); return 1;
    }
    default:
        return 0;
    }
}

}  // namespace alaz
