// host_capi.cpp — flat C entry points over the C++ host side, for the Python test harness and
// for embedding.  `sgh_graphds_create(NULL, ...)` wires GraphDS to a recording stand-in of the
// engine API (host-logic tests, no GPU); a library path wires it to the real libservicegraph.so.
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "graph_ds.hpp"
#include "sockline.hpp"

using namespace alaz;

namespace {

// ---- recording stand-in for the engine (host-logic tests only; computes nothing) ----
struct MockEngine {
    std::vector<sg_event> events;
    std::vector<std::array<uint32_t, 3>> table_ops;   // {op: 1 upsert_pod 2 delete_pod 3 upsert_svc 4 delete_svc, ip, id}
    uint32_t label_count = 0; uint32_t flushes = 0;
};
int m_create(const sg_config*, sg_handle* out) { *out = reinterpret_cast<sg_handle>(new MockEngine()); return SG_OK; }
int m_destroy(sg_handle h) { delete reinterpret_cast<MockEngine*>(h); return SG_OK; }
int m_upsert_pod(sg_handle h, uint32_t ip, uint32_t id) { reinterpret_cast<MockEngine*>(h)->table_ops.push_back({1u, ip, id}); return SG_OK; }
int m_delete_pod(sg_handle h, uint32_t ip) { reinterpret_cast<MockEngine*>(h)->table_ops.push_back({2u, ip, 0u}); return SG_OK; }
int m_upsert_svc(sg_handle h, uint32_t ip, uint32_t id) { reinterpret_cast<MockEngine*>(h)->table_ops.push_back({3u, ip, id}); return SG_OK; }
int m_delete_svc(sg_handle h, uint32_t ip) { reinterpret_cast<MockEngine*>(h)->table_ops.push_back({4u, ip, 0u}); return SG_OK; }
int m_labels(sg_handle h, uint32_t n) { reinterpret_cast<MockEngine*>(h)->label_count = n; return SG_OK; }
int m_ingest(sg_handle h, const sg_event* ev, size_t n) { auto* m = reinterpret_cast<MockEngine*>(h); m->events.insert(m->events.end(), ev, ev + n); return SG_OK; }
int m_flush(sg_handle h, uint64_t, sg_edge_out*, size_t, size_t* n) { reinterpret_cast<MockEngine*>(h)->flushes++; if (n) *n = 0; return SG_OK; }
int m_obips(sg_handle, uint32_t*, size_t, size_t* n) { if (n) *n = 0; return SG_OK; }
const char* m_err(sg_handle) { return ""; }

struct CollectSink : EdgeSink {
    std::vector<EdgeRow> rows; int64_t window_end = 0;
    int PersistEdges(int64_t w, const std::vector<EdgeRow>& r) override { window_end = w; rows = r; return 0; }
};

struct HostCtx {
    void* dl = nullptr; SgApi api; sg_handle h = nullptr; bool mock = false;
    datastore::NullDataStore inner; CollectSink sink;
    std::unique_ptr<GraphDS> ds;
    ConnTracker conns;                 // f-2: TCP connect events -> socket lines -> alive connections
};

}  // namespace

extern "C" {

struct sgh_edge_row {
    char from_type[12], to_type[12], from_uid[160], to_uid[160];
    uint32_t count, err_count; uint64_t sum_ns, max_ns, sumsq_us; float score, lat_z, err_ratio; uint32_t alive;
};

void* sgh_packer_create(void) { return new L7Packer(); }
void sgh_packer_destroy(void* p) { delete static_cast<L7Packer*>(p); }
void sgh_packer_known_ip(void* p, uint32_t ip, int add) { if (add) static_cast<L7Packer*>(p)->AddKnownIP(ip); else static_cast<L7Packer*>(p)->RemoveKnownIP(ip); }
size_t sgh_packer_pack_wire(void* p, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs, sg_event* out, size_t cap) {
    auto* pk = static_cast<L7Packer*>(p);
    std::vector<sg_event> v; l7_req::L7Event e;
    for (size_t i = 0; i < n; i++) { l7_req::DecodeWire(recs + i * l7_req::kWireSize, &e); pk->Pack(e, kafka_msgs ? kafka_msgs[i] : 1u, &v); }
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

void* sgh_graphds_create(const char* engine_lib, const sg_config* cfg, size_t batch) {
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
    c->ds = std::make_unique<GraphDS>(&c->inner, c->api, c->h, &c->sink, cfg ? (size_t)cfg->max_edges : 1024, batch ? batch : 4096);
    return c.release();
}
void sgh_graphds_destroy(void* g) { auto* c = static_cast<HostCtx*>(g); if (!c) return; c->ds.reset(); if (c->h) c->api.destroy(c->h); delete c; }
int sgh_graphds_persist_pod(void* g, const char* et, const char* uid, const char* ip) { datastore::Pod p; p.UID = uid; p.IP = ip; return static_cast<HostCtx*>(g)->ds->PersistPod(p, et); }
int sgh_graphds_persist_service(void* g, const char* et, const char* uid, const char* ip) { datastore::Service s; s.UID = uid; if (ip && *ip) s.ClusterIPs.push_back(ip); return static_cast<HostCtx*>(g)->ds->PersistService(s, et); }
int sgh_graphds_ingest_wire(void* g, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs) {
    auto* c = static_cast<HostCtx*>(g); l7_req::L7Event e; int rc = 0;
    for (size_t i = 0; i < n; i++) { l7_req::DecodeWire(recs + i * l7_req::kWireSize, &e); const int r = c->ds->IngestL7(e, kafka_msgs ? kafka_msgs[i] : 1u); if (r) rc = r; }
    return rc;
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
        o.count = r.Count; o.err_count = r.ErrCount; o.sum_ns = r.SumNs; o.max_ns = r.MaxNs; o.sumsq_us = r.SumSqUs;
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
    for (size_t i = 0; i < n; i++) added += c->conns.ProcessTcpConnect(tcp_state::DecodeWire(recs + i * tcp_state::kWireSize)) ? 1 : 0;
    return added;
}
size_t sgh_graphds_socklines(void* g) { return static_cast<HostCtx*>(g)->conns.Lines(); }
void* sgh_graphds_sockline(void* g, uint32_t pid, uint64_t fd) { return static_cast<HostCtx*>(g)->conns.Line(pid, fd); }
// one clearSocketLines tick: open connections -> GraphDS::PersistAliveConnection (-> SG_EV_ALIVE records)
size_t sgh_graphds_sweep(void* g, int64_t now_ms, int send_alive) { auto* c = static_cast<HostCtx*>(g); return c->conns.Sweep(now_ms, send_alive != 0, c->ds.get()); }

size_t sgh_graphds_labels(void* g, char* buf, size_t cap) { return join_labels(static_cast<HostCtx*>(g)->ds->Labels(), buf, cap); }
uint64_t sgh_graphds_dropped_parse(void* g) { return static_cast<HostCtx*>(g)->ds->Packer().DroppedParse(); }
void* sgh_graphds_engine(void* g) { return static_cast<HostCtx*>(g)->h; }
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
