// host_capi.cpp — flat C entry points over the C++ host side, for the Python test harness and
// for embedding.  `sgh_graphds_create(NULL, ...)` wires GraphDS to a recording stand-in of the
// engine API (host-logic tests, no GPU); a library path wires it to the real libservicegraph.so.
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "graph_ds.hpp"
#include "edges_payload.hpp"
#include "sockline.hpp"

using namespace alaz;

namespace {

// ---- recording stand-in for the engine (host-logic tests only; computes nothing) ----
struct MockEngine {
    std::vector<sg_event> events;
    std::vector<std::array<uint32_t, 3>> table_ops;   // {op: 1 upsert_pod 2 delete_pod 3 upsert_svc 4 delete_svc, ip, id}
    uint32_t label_count = 0; uint32_t flushes = 0; uint32_t max_known = 0x3FFFFFFFu;
    std::mutex mu;                                     // like the real engine, the stand-in serialises the calls on one handle
};
#define M_LOCK(h) std::lock_guard<std::mutex> _g(reinterpret_cast<MockEngine*>(h)->mu)
int m_create(const sg_config* cfg, sg_handle* out) { auto* m = new MockEngine(); if (cfg && cfg->max_known_nodes) m->max_known = cfg->max_known_nodes; *out = reinterpret_cast<sg_handle>(m); return SG_OK; }
int m_destroy(sg_handle h) { delete reinterpret_cast<MockEngine*>(h); return SG_OK; }
int m_upsert_pod(sg_handle h, uint32_t ip, uint32_t id) { M_LOCK(h); auto* m = reinterpret_cast<MockEngine*>(h); if (id >= m->max_known) return SG_ENOSPC; m->table_ops.push_back({1u, ip, id}); return SG_OK; }
int m_delete_pod(sg_handle h, uint32_t ip) { M_LOCK(h); reinterpret_cast<MockEngine*>(h)->table_ops.push_back({2u, ip, 0u}); return SG_OK; }
int m_upsert_svc(sg_handle h, uint32_t ip, uint32_t id) { M_LOCK(h); auto* m = reinterpret_cast<MockEngine*>(h); if (id >= m->max_known) return SG_ENOSPC; m->table_ops.push_back({3u, ip, id}); return SG_OK; }
int m_delete_svc(sg_handle h, uint32_t ip) { M_LOCK(h); reinterpret_cast<MockEngine*>(h)->table_ops.push_back({4u, ip, 0u}); return SG_OK; }
int m_labels(sg_handle h, uint32_t n) { M_LOCK(h); reinterpret_cast<MockEngine*>(h)->label_count = n; return SG_OK; }
int m_ingest(sg_handle h, const sg_event* ev, size_t n) { M_LOCK(h); auto* m = reinterpret_cast<MockEngine*>(h); m->events.insert(m->events.end(), ev, ev + n); return SG_OK; }
int m_flush(sg_handle h, uint64_t, sg_edge_out*, size_t, size_t* n) { M_LOCK(h); reinterpret_cast<MockEngine*>(h)->flushes++; if (n) *n = 0; return SG_OK; }
int m_obips(sg_handle, uint32_t*, size_t, size_t* n) { if (n) *n = 0; return SG_OK; }
const char* m_err(sg_handle) { return ""; }

struct CollectSink : EdgeSink {
    std::vector<EdgeRow> rows; int64_t window_end = 0;
    int PersistEdges(int64_t w, const std::vector<EdgeRow>& r) override { window_end = w; rows = r; return 0; }
};

// the inner data store of the tests: the out-of-scope BackendDS stand-in, counting what reaches it
struct CountingDataStore : datastore::NullDataStore {
    std::atomic<uint64_t> requests{0}, kafka{0}, alive{0}, pods{0}, services{0};
    int PersistPod(const datastore::Pod&, const std::string&) override { pods++; return 0; }
    int PersistService(const datastore::Service&, const std::string&) override { services++; return 0; }
    int PersistRequest(const datastore::Request*) override { requests++; return 0; }
    int PersistKafkaEvent(const datastore::KafkaEvent*) override { kafka++; return 0; }
    int PersistAliveConnection(const datastore::AliveConnection*) override { alive++; return 0; }
};

struct HostCtx {
    void* dl = nullptr; SgApi api; sg_handle h = nullptr; bool mock = false;
    CountingDataStore inner; CollectSink sink;
    std::unique_ptr<GraphDS> ds;
    ConnTracker conns;                 // f-2: TCP connect events -> socket lines -> alive connections
};

}  // namespace

extern "C" {

struct sgh_edge_row {
    char from_type[12], to_type[12], from_uid[160], to_uid[160];
    uint32_t count, err_count; uint64_t sum_ns, max_ns, sumsq_us; float score, lat_z, err_ratio; uint32_t alive; uint32_t p50_us, p99_us;
};

void* sgh_packer_create(void) { return new L7Packer(); }
void sgh_packer_destroy(void* p) { delete static_cast<L7Packer*>(p); }
void sgh_packer_known_ip(void* p, uint32_t ip, int add) { static_cast<L7Packer*>(p)->SetPodIP(ip, add != 0); }
size_t sgh_packer_pack_wire(void* p, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs, sg_event* out, size_t cap) {
    auto* pk = static_cast<L7Packer*>(p);
    std::vector<sg_event> v; v.reserve(n);
    for (size_t i = 0; i < n; i++) pk->PackWire(recs + i * l7_req::kWireSize, kafka_msgs ? kafka_msgs[i] : 1u, &v);
    const size_t m = std::min(cap, v.size());
    if (m) std::memcpy(out, v.data(), m * sizeof(sg_event));
    return v.size();
}
// the same through the user-space L7Event (DecodeWire with the full 1 KiB copy, then L7Packer::Pack): what a caller holding
// l7_req.L7Event objects gets; must equal sgh_packer_pack_wire record for record (tests)
size_t sgh_packer_pack_wire_full(void* p, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs, sg_event* out, size_t cap) {
    auto* pk = static_cast<L7Packer*>(p);
    std::vector<sg_event> v; v.reserve(n);
    auto e = std::make_unique<l7_req::L7Event>();
    for (size_t i = 0; i < n; i++) { l7_req::DecodeWire(recs + i * l7_req::kWireSize, e.get(), true); pk->Pack(*e, kafka_msgs ? kafka_msgs[i] : 1u, &v); }
    const size_t m = std::min(cap, v.size());
    if (m) std::memcpy(out, v.data(), m * sizeof(sg_event));
    return v.size();
}
static size_t join_labels(const std::vector<std::string>& l, char* buf, size_t cap) {
    std::string s;
    for (size_t i = 0; i < l.size(); i++) { if (i) s.push_back('\n'); s += l[i]; }
    if (buf && cap) { const size_t m = std::min(cap - 1, s.size()); std::memcpy(buf, s.data(), m); buf[m] = 0; }
    return l.size();
}
size_t sgh_packer_labels(void* p, char* buf, size_t cap) { return join_labels(static_cast<L7Packer*>(p)->Labels(), buf, cap); }
uint64_t sgh_packer_dropped_parse(void* p) { return static_cast<L7Packer*>(p)->DroppedParse(); }
void sgh_parse_http(const char* req, size_t len, char* m, char* p, char* v, char* h, size_t cap) {
    std::string sm, sp, sv, sh;
    ParseHttpPayload(req, len, &sm, &sp, &sv, &sh);
    auto put = [&](char* d, const std::string& s) { const size_t k = std::min(cap - 1, s.size()); std::memcpy(d, s.data(), k); d[k] = 0; };
    put(m, sm); put(p, sp); put(v, sv); put(h, sh);
}

void* sgh_graphds_create2(const char* engine_lib, const sg_config* cfg, size_t batch, int divert_requests);
void* sgh_graphds_create(const char* engine_lib, const sg_config* cfg, size_t batch) { return sgh_graphds_create2(engine_lib, cfg, batch, 0); }
void* sgh_graphds_create2(const char* engine_lib, const sg_config* cfg, size_t batch, int divert_requests) {
    auto c = std::make_unique<HostCtx>();
    if (engine_lib) {
        c->dl = dlopen(engine_lib, RTLD_NOW | RTLD_GLOBAL);
        if (!c->dl || !SgApi::FromLibrary(c->dl, &c->api)) return nullptr;
    } else {
        c->mock = true;
        c->api.create = m_create; c->api.destroy = m_destroy; c->api.upsert_pod = m_upsert_pod; c->api.delete_pod = m_delete_pod;
        c->api.upsert_service = m_upsert_svc; c->api.delete_service = m_delete_svc; c->api.set_label_count = m_labels; c->api.ingest = m_ingest;
        c->api.flush_window = m_flush; c->api.window_outbound_ips = m_obips; c->api.last_error = m_err;
    }
    if (c->api.create(cfg, &c->h) != SG_OK) return nullptr;        // no usable GPU => no GraphDS: there is no CPU fallback
    c->ds = std::make_unique<GraphDS>(&c->inner, c->api, c->h, &c->sink, cfg ? (size_t)cfg->max_edges : 1024, batch ? batch : 4096,
                                      cfg && cfg->max_known_nodes ? cfg->max_known_nodes : 0x3FFFFFFFu, divert_requests != 0);
    return c.release();
}
void sgh_graphds_destroy(void* g) { auto* c = static_cast<HostCtx*>(g); if (!c) return; c->ds.reset(); if (c->h) c->api.destroy(c->h); delete c; }
int sgh_graphds_persist_pod(void* g, const char* et, const char* uid, const char* ip) { datastore::Pod p; p.UID = uid; p.IP = ip; return static_cast<HostCtx*>(g)->ds->PersistPod(p, et); }
int sgh_graphds_persist_service(void* g, const char* et, const char* uid, const char* ip) { datastore::Service s; s.UID = uid; if (ip && *ip) s.ClusterIPs.push_back(ip); return static_cast<HostCtx*>(g)->ds->PersistService(s, et); }
int sgh_graphds_ingest_wire(void* g, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs) {
    return static_cast<HostCtx*>(g)->ds->IngestWire(recs, n, kafka_msgs);
}
int sgh_graphds_persist_request(void* g, int64_t start_ms, uint64_t latency, const char* from_ip, const char* from_type, const char* from_uid,
                                const char* to_ip, const char* to_type, const char* to_uid, const char* protocol, uint32_t status,
                                const char* method, int tls) {
    datastore::Request r; r.StartTime = start_ms; r.Latency = latency; r.FromIP = from_ip; r.FromType = from_type; r.FromUID = from_uid;
    r.ToIP = to_ip; r.ToType = to_type; r.ToUID = to_uid; r.Protocol = protocol; r.StatusCode = status; r.Method = method; r.Tls = tls != 0;
    return static_cast<HostCtx*>(g)->ds->PersistRequest(&r);
}
long sgh_graphds_flush(void* g, int64_t window_end_ms, sgh_edge_row* out, size_t cap) {
    auto* c = static_cast<HostCtx*>(g);
    const long n = c->ds->FlushWindow(window_end_ms);
    if (n < 0) return n;
    const size_t m = std::min(cap, c->sink.rows.size());
    for (size_t i = 0; i < m; i++) {
        const EdgeRow& r = c->sink.rows[i]; sgh_edge_row& o = out[i];
        std::memset(&o, 0, sizeof o);
        std::strncpy(o.from_type, r.FromType.c_str(), sizeof o.from_type - 1); std::strncpy(o.to_type, r.ToType.c_str(), sizeof o.to_type - 1);
        std::strncpy(o.from_uid, r.FromUID.c_str(), sizeof o.from_uid - 1); std::strncpy(o.to_uid, r.ToUID.c_str(), sizeof o.to_uid - 1);
        o.count = r.Count; o.err_count = r.ErrCount; o.sum_ns = r.SumNs; o.max_ns = r.MaxNs; o.sumsq_us = r.SumSqUs; o.p50_us = r.P50Us; o.p99_us = r.P99Us;
        o.score = r.Score; o.lat_z = r.LatZ; o.err_ratio = r.ErrRatio; o.alive = r.Alive;
    }
    return n;
}
// ---- f-2: socket lines ----
struct sgh_sockinfo { uint32_t pid; uint64_t fd; uint32_t saddr; uint16_t sport; uint32_t daddr; uint16_t dport; };
void* sgh_sockline_create(uint32_t pid, uint64_t fd) { return new SocketLine(pid, fd); }
void sgh_sockline_destroy(void* l) { delete static_cast<SocketLine*>(l); }
void sgh_sockline_add(void* l, uint64_t ts, const sgh_sockinfo* si) {
    if (!si) { static_cast<SocketLine*>(l)->AddValue(ts, nullptr); return; }
    SockInfo s; s.Pid = si->pid; s.Fd = si->fd; s.Saddr = si->saddr; s.Sport = si->sport; s.Daddr = si->daddr; s.Dport = si->dport;
    static_cast<SocketLine*>(l)->AddValue(ts, &s);
}
int sgh_sockline_get(void* l, uint64_t ts, uint64_t now_ns, sgh_sockinfo* out) {
    SockInfo s; const SockErr e = static_cast<SocketLine*>(l)->GetValue(ts, now_ns, &s);
    if (e == SockErr::Ok && out) { out->pid = s.Pid; out->fd = s.Fd; out->saddr = s.Saddr; out->sport = s.Sport; out->daddr = s.Daddr; out->dport = s.Dport; }
    return (int)e;
}
// getConnectionInfo against `proc_root` (sock_num_line.go:399-429): 0 = seeded, else SocketLine::Seed
int sgh_sockline_seed(void* l, const char* proc_root, uint64_t now_kernel_ns) { return (int)static_cast<SocketLine*>(l)->SeedFromProc(proc_root, now_kernel_ns); }
int sgh_proc_inode_of_link(const char* link, char* inode, size_t cap) {
    std::string v;
    if (!procfs::InodeOfLink(link, &v) || v.size() + 1 > cap) return -1;
    std::memcpy(inode, v.c_str(), v.size() + 1);
    return 0;
}
int sgh_proc_parse_tcp_line(const char* line, uint32_t* laddr, uint16_t* lport, uint32_t* raddr, uint16_t* rport) {
    return procfs::ParseTcpLine(line, laddr, lport, raddr, rport) ? 0 : -1;
}
void sgh_sockline_delete_unused(void* l) { static_cast<SocketLine*>(l)->DeleteUnused(); }
size_t sgh_sockline_len(void* l) { return static_cast<SocketLine*>(l)->Size(); }
int sgh_sockline_at(void* l, size_t i, uint64_t* ts, uint64_t* last_match, sgh_sockinfo* out) {
    auto* sl = static_cast<SocketLine*>(l);
    if (i >= sl->Size()) return -1;
    const TimestampedSocket v = sl->At(i);
    if (ts) *ts = v.Timestamp;
    if (last_match) *last_match = v.LastMatch;
    if (out && v.Open) { out->pid = v.Info.Pid; out->fd = v.Info.Fd; out->saddr = v.Info.Saddr; out->sport = v.Info.Sport; out->daddr = v.Info.Daddr; out->dport = v.Info.Dport; }
    return v.Open ? 1 : 0;
}
// TCP connect records (BpfTcpEvent, 64 bytes each) through processTcpConnect; returns values added
size_t sgh_graphds_tcp_wire(void* g, const uint8_t* recs, size_t n) {
    auto* c = static_cast<HostCtx*>(g); size_t added = 0;
    for (size_t i = 0; i < n; i++) {
        const tcp_state::TcpConnectEvent e = tcp_state::DecodeWire(recs + i * tcp_state::kWireSize);
        const bool ok = c->conns.ProcessTcpConnect(e);
        if (ok && e.Type == tcp_state::kClosed) c->ds->ConnClosed(e.Pid, e.Fd);     // data.go:484-494
        added += ok ? 1 : 0;
    }
    return added;
}
// process exec / exit (proc events, data.go:354-377) and the HTTP/2 minute sweep (:553-567)
void sgh_graphds_proc_exec(void* g, uint32_t pid) { static_cast<HostCtx*>(g)->ds->ProcExec(pid); }
void sgh_graphds_proc_exit(void* g, uint32_t pid) { auto* c = static_cast<HostCtx*>(g); c->conns.ClearProc(pid); c->ds->ProcExit(pid); }   // clearProc first, data.go:364
// new socket lines are seeded from `root` (NULL or "" = off); the seeded value is stamped first_kernel − (first_user − now_user)
void sgh_graphds_set_proc_root(void* g, const char* root, uint64_t first_kernel_ns, uint64_t first_user_ns, uint64_t now_user_ns) {
    static_cast<HostCtx*>(g)->conns.SetProcRoot(root ? root : "", first_kernel_ns, first_user_ns, now_user_ns);
}
size_t sgh_graphds_pg_statements(void* g) { return static_cast<HostCtx*>(g)->ds->Packer().PgStatements(); }
void sgh_graphds_seed_stats(void* g, uint64_t out[2]) { auto* c = static_cast<HostCtx*>(g); out[0] = c->conns.SeedsOk(); out[1] = c->conns.SeedsFailed(); }
void sgh_graphds_sweep_http2(void* g) { static_cast<HostCtx*>(g)->ds->SweepHttp2(); }
// counters of the assembler: [0] streams pending, [1] parsers, [2] dropped: pid not live, [3] dropped: method/path unparsed, [4] dropped: time
void sgh_graphds_http2_stats(void* g, uint64_t out[5]) {
    const Http2Assembler& a = static_cast<HostCtx*>(g)->ds->Packer().Http2();
    out[0] = a.Pending(); out[1] = a.Parsers(); out[2] = a.DroppedNotLive(); out[3] = a.DroppedUnparsed(); out[4] = a.DroppedTime();
}

// ---- stand-alone HPACK decoder / HTTP/2 assembler (known-answer and differential tests) ----
void* sgh_hpack_create(uint32_t max_table) { return new hpack::Decoder(max_table); }
void sgh_hpack_destroy(void* d) { delete static_cast<hpack::Decoder*>(d); }
// decodes `block`; the emitted fields are appended to buf as name\0value\0 (lengths in lens[2*i], lens[2*i+1]).
// returns the number of fields, or -(1 + fields emitted before the error) on a decoding error
long sgh_hpack_write(void* d, const uint8_t* block, size_t n, char* buf, size_t cap, uint32_t* lens, size_t max_fields) {
    auto* dec = static_cast<hpack::Decoder*>(d);
    size_t used = 0, nf = 0;
    dec->SetEmitFunc([&](std::string_view name, std::string_view value) {
        if (nf < max_fields && used + name.size() + value.size() <= cap) {
            std::memcpy(buf + used, name.data(), name.size()); used += name.size();
            std::memcpy(buf + used, value.data(), value.size()); used += value.size();
            lens[2 * nf] = (uint32_t)name.size(); lens[2 * nf + 1] = (uint32_t)value.size();
        }
        nf++;
    });
    const bool ok = dec->Write(block, n);
    dec->SetEmitFunc(nullptr);
    return ok ? (long)nf : -(long)(1 + nf);
}
size_t sgh_hpack_table_len(void* d) { return static_cast<hpack::Decoder*>(d)->DynamicTableLen(); }
uint32_t sgh_hpack_table_size(void* d) { return static_cast<hpack::Decoder*>(d)->DynamicTableSize(); }
size_t sgh_hpack_table_at(void* d, size_t i, char* buf, size_t cap, uint32_t lens[2]) {
    const hpack::HeaderField& f = static_cast<hpack::Decoder*>(d)->DynamicTableAt(i);
    lens[0] = (uint32_t)f.Name.size(); lens[1] = (uint32_t)f.Value.size();
    if (f.Name.size() + f.Value.size() > cap) return 0;
    std::memcpy(buf, f.Name.data(), f.Name.size()); std::memcpy(buf + f.Name.size(), f.Value.data(), f.Value.size());
    return f.Name.size() + f.Value.size();
}
long sgh_huffman_decode(const uint8_t* p, size_t n, char* out, size_t cap) {
    std::string s; if (!hpack::HuffmanDecode(p, n, &s)) return -1;
    if (s.size() > cap) return -2;
    std::memcpy(out, s.data(), s.size()); return (long)s.size();
}
long sgh_huffman_encode(const char* p, size_t n, char* out, size_t cap) {
    std::string s; hpack::HuffmanEncode(std::string(p, n), &s);
    if (s.size() > cap) return -2;
    std::memcpy(out, s.data(), s.size()); return (long)s.size();
}
uint32_t sgh_go_atoi_u32(const char* p, size_t n) { return GoAtoiU32(std::string_view(p, n)); }

struct sgh_h2_out { char method[64]; char path[1100]; char authority[160]; char protocol[8]; uint32_t status_code; uint64_t latency; };
void* sgh_h2_create(void) { return new Http2Assembler(); }
void sgh_h2_destroy(void* a) { delete static_cast<Http2Assembler*>(a); }
int sgh_h2_event(void* a, uint32_t pid, uint64_t fd, int method_id, const uint8_t* payload, uint32_t size, uint64_t write_ns, int tls, sgh_h2_out* out) {
    l7_req::L7Event e; e.Pid = pid; e.Fd = fd; e.ProtocolId = l7_req::BPF_L7_PROTOCOL_HTTP2; e.MethodId = (uint8_t)method_id;
    e.PayloadSize = size > l7_req::kMaxPayload ? (uint32_t)l7_req::kMaxPayload : size; std::memcpy(e.Payload, payload, e.PayloadSize);
    e.WriteTimeNs = write_ns; e.Tls = tls != 0;
    Http2Request r;
    if (!static_cast<Http2Assembler*>(a)->OnEvent(e, &r)) return 0;
    auto put = [](char* dst, size_t cap, const std::string& s) { const size_t n = s.size() < cap - 1 ? s.size() : cap - 1; std::memcpy(dst, s.data(), n); dst[n] = 0; };
    std::memset(out, 0, sizeof *out);
    put(out->method, sizeof out->method, r.Method); put(out->path, sizeof out->path, r.Path); put(out->authority, sizeof out->authority, r.Authority);
    put(out->protocol, sizeof out->protocol, r.Protocol); out->status_code = r.StatusCode; out->latency = r.Latency;
    return 1;
}
void sgh_h2_proc_exec(void* a, uint32_t pid) { static_cast<Http2Assembler*>(a)->ProcExec(pid); }
void sgh_h2_proc_exit(void* a, uint32_t pid) { static_cast<Http2Assembler*>(a)->ProcExit(pid); }
void sgh_h2_conn_closed(void* a, uint32_t pid, uint64_t fd) { static_cast<Http2Assembler*>(a)->ConnClosed(pid, fd); }
void sgh_h2_sweep(void* a) { static_cast<Http2Assembler*>(a)->Sweep(); }
size_t sgh_h2_pending(void* a) { return static_cast<Http2Assembler*>(a)->Pending(); }
size_t sgh_h2_parsers(void* a) { return static_cast<Http2Assembler*>(a)->Parsers(); }
// ---- edge egress (edges_payload.hpp, f-3) ----
// the rows of the last sgh_graphds_flush as "/edges/" payloads of at most `batch` rows, concatenated with '\n' into buf;
// returns the number of payloads (0 when the window had no edge), or -(bytes needed) when buf is too small
long sgh_graphds_edges_json(void* g, const char* monitoring_id, const char* idem_key, const char* node_id, const char* version, size_t batch,
                            char* buf, size_t cap) {
    auto* c = static_cast<HostCtx*>(g);
    std::string all; long n = 0;
    JsonEdgeSink sink(PayloadMetadata{monitoring_id ? monitoring_id : "", idem_key ? idem_key : "", node_id ? node_id : "", version ? version : ""}, batch,
                      [&](const char*, const std::string& body) { if (n) all.push_back('\n'); all += body; n++; return 0; });
    sink.PersistEdges(c->sink.window_end, c->sink.rows);
    if (all.size() + 1 > cap) return -(long)(all.size() + 1);
    std::memcpy(buf, all.data(), all.size()); buf[all.size()] = 0;
    return n;
}
// the same encoder over caller-supplied rows (tests)
long sgh_edges_json_from_rows(const sgh_edge_row* in, size_t n_rows, int64_t window_end_ms, const char* monitoring_id, const char* idem_key, const char* node_id,
                              const char* version, size_t batch, char* buf, size_t cap) {
    std::vector<EdgeRow> rows(n_rows);
    for (size_t i = 0; i < n_rows; i++) {
        EdgeRow& r = rows[i]; const sgh_edge_row& o = in[i];
        r.FromType = o.from_type; r.FromUID = o.from_uid; r.ToType = o.to_type; r.ToUID = o.to_uid;
        r.Count = o.count; r.ErrCount = o.err_count; r.SumNs = o.sum_ns; r.MaxNs = o.max_ns; r.SumSqUs = o.sumsq_us; r.P50Us = o.p50_us; r.P99Us = o.p99_us;
        r.Score = o.score; r.LatZ = o.lat_z; r.ErrRatio = o.err_ratio; r.Alive = o.alive;
    }
    std::string all; long n = 0;
    JsonEdgeSink sink(PayloadMetadata{monitoring_id ? monitoring_id : "", idem_key ? idem_key : "", node_id ? node_id : "", version ? version : ""}, batch,
                      [&](const char*, const std::string& body) { if (n) all.push_back('\n'); all += body; n++; return 0; });
    sink.PersistEdges(window_end_ms, rows);
    if (all.size() + 1 > cap) return -(long)(all.size() + 1);
    std::memcpy(buf, all.data(), all.size()); buf[all.size()] = 0;
    return n;
}
size_t sgh_json_string(const char* s, size_t n, char* out, size_t cap) {
    std::string o; AppendJsonString(std::string(s, n), &o);
    if (o.size() + 1 <= cap) { std::memcpy(out, o.data(), o.size()); out[o.size()] = 0; }
    return o.size();
}

// ---- Kafka payload decode (kafka.hpp) ----
void sgh_packer_kafka_decode(void* p, int on) { static_cast<L7Packer*>(p)->SetKafkaDecode(on != 0); }
void sgh_graphds_kafka_decode(void* g, int on) { static_cast<HostCtx*>(g)->ds->SetKafkaDecode(on != 0); }
// decodes one payload; messages are serialised into buf as [u32 topic_n][i32 partition][u32 key_n][u32 value_n] topic key value ...
// returns the message count (also when buf is too small: then nothing is written), *status = kafka::Status
long sgh_kafka_decode(const uint8_t* payload, size_t size, int method_id, int api_version, int* status, uint8_t* buf, size_t cap) {
    std::vector<kafka::Message> msgs; size_t counted = 0, counted_only = 0;
    const kafka::Status st = kafka::DecodePayload(payload, size, method_id, (int16_t)api_version, &msgs, &counted);
    const kafka::Status st2 = kafka::DecodePayload(payload, size, method_id, (int16_t)api_version, nullptr, &counted_only);   // the packer's mode
    if (st2 != st || counted != msgs.size() || counted_only != counted) { if (status) *status = 99; return -1; }           // self-check of the two modes
    if (status) *status = (int)st;
    size_t need = 0;
    for (const auto& m : msgs) need += 16 + m.Topic.size() + m.Key.size() + m.Value.size();
    if (need <= cap) {
        uint8_t* w = buf;
        for (const auto& m : msgs) {
            const uint32_t h[4] = {(uint32_t)m.Topic.size(), (uint32_t)m.Partition, (uint32_t)m.Key.size(), (uint32_t)m.Value.size()};
            std::memcpy(w, h, 16); w += 16;
            std::memcpy(w, m.Topic.data(), m.Topic.size()); w += m.Topic.size();
            std::memcpy(w, m.Key.data(), m.Key.size()); w += m.Key.size();
            std::memcpy(w, m.Value.data(), m.Value.size()); w += m.Value.size();
        }
    }
    return (long)msgs.size();
}
long sgh_kafka_decompress(int codec, const uint8_t* src, size_t n, uint8_t* out, size_t cap) {
    std::string s; if (!kafka::Decompress(codec, src, n, &s)) return -1;
    if (s.size() > cap) return -2;
    std::memcpy(out, s.data(), s.size()); return (long)s.size();
}
uint32_t sgh_crc32(int castagnoli, const uint8_t* p, size_t n) { return kafka::Crc32(p, n, castagnoli != 0); }
uint32_t sgh_xxh32(const uint8_t* p, size_t n, uint32_t seed) { return kafka::XXH32(p, n, seed); }
// the stand-alone packer's assembler
void sgh_packer_proc_exec(void* p, uint32_t pid) { static_cast<L7Packer*>(p)->Http2().ProcExec(pid); }
void sgh_packer_proc_exit(void* p, uint32_t pid) { static_cast<L7Packer*>(p)->ProcExit(pid); }
void sgh_packer_conn_closed(void* p, uint32_t pid, uint64_t fd) { static_cast<L7Packer*>(p)->ConnClosed(pid, fd); }
size_t sgh_packer_pg_statements(void* p) { return static_cast<L7Packer*>(p)->PgStatements(); }
size_t sgh_graphds_socklines(void* g) { return static_cast<HostCtx*>(g)->conns.Lines(); }
// An OWNING reference to the line of (pid, fd), or null: sgh_graphds_proc_exit -> ClearProc(pid) drops the tracker's reference, and a
// wrapper that still looks at the line must not be left with freed memory.  sgh_sockline_ref_get is the SocketLine* the
// sgh_sockline_* calls take, valid until sgh_sockline_ref_release.
void* sgh_graphds_sockline(void* g, uint32_t pid, uint64_t fd) {
    auto sp = static_cast<HostCtx*>(g)->conns.Share(pid, fd);
    return sp ? new std::shared_ptr<SocketLine>(std::move(sp)) : nullptr;
}
void* sgh_sockline_ref_get(void* ref) { return ref ? static_cast<std::shared_ptr<SocketLine>*>(ref)->get() : nullptr; }
void sgh_sockline_ref_release(void* ref) { delete static_cast<std::shared_ptr<SocketLine>*>(ref); }
// one clearSocketLines tick: open connections -> GraphDS::PersistAliveConnection (-> SG_EV_ALIVE records)
size_t sgh_graphds_sweep(void* g, int64_t now_ms, int send_alive) { auto* c = static_cast<HostCtx*>(g); return c->conns.Sweep(now_ms, send_alive != 0, c->ds.get()); }

size_t sgh_graphds_labels(void* g, char* buf, size_t cap) { return join_labels(static_cast<HostCtx*>(g)->ds->Labels(), buf, cap); }
uint64_t sgh_graphds_dropped_parse(void* g) { return static_cast<HostCtx*>(g)->ds->Packer().DroppedParse(); }
void* sgh_graphds_engine(void* g) { return static_cast<HostCtx*>(g)->h; }
// {events offered, batches dropped, engine errors, live node ids, requests / kafka events / alive connections / pods / services that reached the inner store}
void sgh_graphds_counters(void* g, uint64_t out[9]) {
    auto* c = static_cast<HostCtx*>(g);
    out[0] = c->ds->EventsOffered(); out[1] = c->ds->BatchesDropped(); out[2] = c->ds->EngineErrors(); out[3] = c->ds->LiveIds();
    out[4] = c->inner.requests; out[5] = c->inner.kafka; out[6] = c->inner.alive; out[7] = c->inner.pods; out[8] = c->inner.services;
}
// mock inspection
size_t sgh_mock_events(void* g, sg_event* out, size_t cap) {
    auto* c = static_cast<HostCtx*>(g); if (!c->mock) return 0;
    auto* m = reinterpret_cast<MockEngine*>(c->h);
    const size_t k = std::min(cap, m->events.size()); if (k) std::memcpy(out, m->events.data(), k * sizeof(sg_event));
    return m->events.size();
}
size_t sgh_mock_table_ops(void* g, uint32_t* out3, size_t cap) {
    auto* c = static_cast<HostCtx*>(g); if (!c->mock) return 0;
    auto* m = reinterpret_cast<MockEngine*>(c->h);
    const size_t k = std::min(cap, m->table_ops.size());
    for (size_t i = 0; i < k; i++) { out3[3 * i] = m->table_ops[i][0]; out3[3 * i + 1] = m->table_ops[i][1]; out3[3 * i + 2] = m->table_ops[i][2]; }
    return m->table_ops.size();
}
uint32_t sgh_mock_label_count(void* g) { auto* c = static_cast<HostCtx*>(g); return c->mock ? reinterpret_cast<MockEngine*>(c->h)->label_count : 0; }

}  // extern "C"

// ---- the sharded window sequence with caller-supplied stages and collectives (alaz_amd/csrc/shard_seq.hpp) ------------------
// The engine library runs the very same function with its HIP stages and RCCL (sg_window_run_sharded); exported here so that
// the sequence can be driven without a GPU: tests/test_sharded.py plugs the numpy stand-in backend and gloo collectives in.
#include "../shard_seq.hpp"
extern "C" int sgh_run_sharded_window(const sg_shard_stages* stages, const sg_shard_comm* comm) { return sg_run_sharded_window(stages, comm); }

// ---- synthetic capture for the streaming harness (tools/c5_stream.py) ---------------------------------------------------------
// BASELINE config 5 is a STREAM of raw l7_event records over a 20 M-edge graph.  A ring of distinct 1096-byte records large enough
// to touch millions of edges per window would be tens of GB; instead the harness keeps a ring of PACKED events (32 bytes each,
// drawn from the seeded edge list by alaz_amd/replay.py) and this feeder expands every one of them into the wire record the
// reference's perf reader would have seen — the same fields and payloads replay.to_wire() writes (ebpf/l7_req/l7.go:345-369;
// "GET /user HTTP1.1" + Host header, a Postgres simple query, count-only Kafka) — in a thread-local buffer, 256 records at a
// time, and hands them to GraphDS::IngestWire: payload parse, label interning, packing and batching all run as in production.
// Paced: the caller's share of `rate` events/s since *t0 (steady_clock ns); returns the events fed, stops when *stop != 0.
#include <atomic>
#include <chrono>
#include <thread>
extern "C" long sgh_graphds_feed_expanded(void* g, const sg_event* ring, size_t ring_n, size_t first, size_t stride_chunks, size_t chunk,
                                          const char* const* labels, size_t n_labels, double rate_per_s, const int64_t* t0_ns, const volatile int* stop,
                                          volatile long* fed_out) {
    auto* ctx = static_cast<HostCtx*>(g);
    constexpr size_t kRec = l7_req::kWireSize, kBatch = 256;
    std::vector<uint8_t> buf(kBatch * kRec);
    static const char kPg[] = "SELECT * FROM users WHERE id = 1";
    long fed = 0;
    size_t pos = first;
    auto now_ns = [] { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (!*stop) {
        const double ahead = (double)fed - rate_per_s * (double)(now_ns() - *t0_ns) * 1e-9;
        if (ahead > 0) { std::this_thread::sleep_for(std::chrono::microseconds((long)std::min(2000.0, ahead / rate_per_s * 1e6 + 1.0))); continue; }
        const size_t n_chunk = std::min(chunk, ring_n - pos);
        for (size_t i0 = 0; i0 < n_chunk; i0 += kBatch) {
            const size_t nb = std::min(kBatch, n_chunk - i0);
            for (size_t i = 0; i < nb; i++) {
                const sg_event& e = ring[pos + i0 + i];
                uint8_t* r = buf.data() + i * kRec;
                std::memset(r, 0, 36); std::memset(r + 1060, 0, kRec - 1060);
                const uint64_t fd = 3 + (pos + i0 + i) % 97; const uint32_t pid = 4242, status = e.status;
                std::memcpy(r, &fd, 8); std::memcpy(r + 8, &e.write_time_ns, 8); std::memcpy(r + 16, &pid, 4); std::memcpy(r + 20, &status, 4);
                std::memcpy(r + 24, &e.duration_ns, 8);
                r[32] = e.protocol;
                uint32_t pl = 0;
                char* p = reinterpret_cast<char*>(r + 36);
                if (e.protocol == SG_PROTO_HTTP) {
                    r[33] = 1;                                                          // GET (ebpf/l7_req/l7.go method enum)
                    const bool external = e.daddr >= 0x08080001u && e.daddr < 0x08090001u;
                    if (e.host_label && e.host_label <= n_labels) pl = (uint32_t)std::snprintf(p, 1024, "GET /user HTTP1.1\r\nHost: %s\r\nAccept: */*\r\n\r\n", labels[e.host_label - 1]);
                    else if (external) pl = (uint32_t)std::snprintf(p, 1024, "GET /user HTTP1.1\r\nAccept: */*\r\n\r\n");
                    else pl = (uint32_t)std::snprintf(p, 1024, "GET /user HTTP1.1\r\nHost: svc.cluster.local\r\nAccept: */*\r\n\r\n");
                } else if (e.protocol == SG_PROTO_POSTGRES) {
                    r[33] = 2;                                                          // SIMPLE_QUERY
                    const uint32_t len = (uint32_t)sizeof kPg + 4;                      // query + NUL + the length word
                    p[0] = 'Q'; p[1] = (char)(len >> 24); p[2] = (char)(len >> 16); p[3] = (char)(len >> 8); p[4] = (char)len;
                    std::memcpy(p + 5, kPg, sizeof kPg); pl = 5 + (uint32_t)sizeof kPg;
                } else if (e.protocol == SG_PROTO_KAFKA) r[33] = (e.flags & SG_EV_CONSUME) ? 2 : 1;
                std::memcpy(r + 1060, &pl, 4); r[1064] = 1; r[1066] = (e.flags & SG_EV_TLS) ? 1 : 0;
                const uint16_t sport = (uint16_t)(32768 + (pos + i0 + i) % 28232), dport = e.protocol == SG_PROTO_POSTGRES ? 5432 : (e.protocol == SG_PROTO_KAFKA ? 9092 : 80);
                std::memcpy(r + 1076, &e.saddr, 4); std::memcpy(r + 1080, &sport, 2); std::memcpy(r + 1084, &e.daddr, 4); std::memcpy(r + 1088, &dport, 2);
            }
            ctx->ds->IngestWire(buf.data(), nb, nullptr);
            fed += (long)nb;
        }
        if (fed_out) *fed_out = fed;
        pos += stride_chunks * chunk;
        if (pos >= ring_n) pos = first;
    }
    if (fed_out) *fed_out = fed;
    return fed;
}
