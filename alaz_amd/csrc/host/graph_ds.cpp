#include "graph_ds.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>

namespace alaz {

bool SgApi::FromLibrary(void* dl, SgApi* o) {
    if (!dl || !o) return false;
#define SG_SYM(field, name) do { o->field = reinterpret_cast<decltype(o->field)>(dlsym(dl, name)); if (!o->field) return false; } while (0)
    SG_SYM(create, "sg_create"); SG_SYM(destroy, "sg_destroy"); SG_SYM(upsert_pod, "sg_upsert_pod"); SG_SYM(delete_pod, "sg_delete_pod");
    SG_SYM(upsert_service, "sg_upsert_service"); SG_SYM(delete_service, "sg_delete_service"); SG_SYM(set_label_count, "sg_set_label_count");
    SG_SYM(ingest, "sg_ingest"); SG_SYM(flush_window, "sg_flush_window"); SG_SYM(window_outbound_ips, "sg_window_outbound_ips");
    SG_SYM(last_error, "sg_last_error");
#undef SG_SYM
    return true;
}

bool ParseIPv4(const std::string& s, uint32_t* out) {
    unsigned a, b, c, d; char tail;
    if (std::sscanf(s.c_str(), "%u.%u.%u.%u%c", &a, &b, &c, &d, &tail) != 4 || a > 255 || b > 255 || c > 255 || d > 255) return false;
    *out = (a << 24) | (b << 16) | (c << 8) | d;
    return true;
}
std::string FormatIPv4(uint32_t ip) {
    char buf[16];
    std::snprintf(buf, sizeof buf, "%u.%u.%u.%u", ip >> 24, (ip >> 16) & 255, (ip >> 8) & 255, ip & 255);
    return buf;
}

GraphDS::GraphDS(datastore::DataStore* inner, const SgApi& api, sg_handle h, EdgeSink* sink, size_t max_edges, size_t batch)
    : inner_(inner), api_(api), h_(h), sink_(sink), max_edges_(max_edges), batch_cap_(batch) { batch_.reserve(batch); }
GraphDS::~GraphDS() = default;

uint32_t GraphDS::Intern(const std::string& uid, uint8_t kind) {
    auto it = ids_.find(uid);
    if (it != ids_.end()) { kind_of_[it->second] = kind; return it->second; }
    const uint32_t id = (uint32_t)uid_of_.size();
    ids_.emplace(uid, id); uid_of_.push_back(uid); kind_of_.push_back(kind);
    return id;
}

// processPod keeps PodIPToPodUid (aggregator/persist.go:55-71); pods without an IP never reach the
// datastore (persist.go:37-40), an empty IP here is ignored for the same reason.
int GraphDS::PersistPod(const datastore::Pod& pod, const std::string& et) {
    uint32_t ip;
    if (!pod.IP.empty() && ParseIPv4(pod.IP, &ip)) {
        std::lock_guard<std::mutex> g(mu_);
        const uint32_t id = Intern(pod.UID, SG_NODE_POD);
        if (et == datastore::ADD || et == datastore::UPDATE) { if (api_.upsert_pod(h_, ip, id) == SG_OK) packer_.AddKnownIP(ip); }
        else if (et == datastore::DELETE_) { api_.delete_pod(h_, ip); packer_.RemoveKnownIP(ip); }
    }
    return inner_->PersistPod(pod, et);
}

// processSvc keys ServiceIPToServiceUid on Spec.ClusterIP (persist.go:114-130); the DTO carries it as
// ClusterIPs[0] (ClusterIP itself is never filled in, persist.go:105-112).
int GraphDS::PersistService(const datastore::Service& svc, const std::string& et) {
    const std::string& ips = !svc.ClusterIPs.empty() ? svc.ClusterIPs[0] : svc.ClusterIP;
    uint32_t ip;
    if (ParseIPv4(ips, &ip)) {
        std::lock_guard<std::mutex> g(mu_);
        const uint32_t id = Intern(svc.UID, SG_NODE_SERVICE);
        if (et == datastore::ADD || et == datastore::UPDATE) { if (api_.upsert_service(h_, ip, id) == SG_OK) packer_.AddKnownIP(ip); }
        else if (et == datastore::DELETE_) { api_.delete_service(h_, ip); packer_.RemoveKnownIP(ip); }
    }
    return inner_->PersistService(svc, et);
}

int GraphDS::FlushBatchLocked() {
    if (batch_.empty()) return SG_OK;
    const int rc = api_.ingest(h_, batch_.data(), batch_.size());
    if (rc == SG_EAGAIN) batches_dropped_++;          // never block the aggregator (the reference would: backend.go:844)
    batch_.clear();
    return rc == SG_EAGAIN ? SG_OK : rc;
}

int GraphDS::Append(const sg_event& ev) {
    batch_.push_back(ev);
    offered_++;
    return batch_.size() >= batch_cap_ ? FlushBatchLocked() : SG_OK;
}

static uint8_t ProtocolId(const std::string& p, bool* tls) {
    if (p == "HTTPS") { *tls = true; return SG_PROTO_HTTP; }     // rewrite of data.go:1240-1242 undone
    static const char* names[] = {"UNKNOWN", "HTTP", "AMQP", "POSTGRES", "HTTP2", "REDIS", "KAFKA", "MYSQL", "MONGO"};
    for (uint8_t i = 0; i <= 8; i++) if (p == names[i]) return i;
    return SG_PROTO_UNKNOWN;
}

// The datastore-boundary tap.  The DTO is only read during the call (cgo rule; backend.go:824-839).
int GraphDS::PersistAliveConnection(const datastore::AliveConnection* c) {
    if (!c) return SG_EINVAL;
    sg_event ev; std::memset(&ev, 0, sizeof ev);
    if (ParseIPv4(c->FromIP, &ev.saddr) && ParseIPv4(c->ToIP, &ev.daddr)) {
        ev.flags = SG_EV_ALIVE;
        std::lock_guard<std::mutex> g(mu_);
        Append(ev);
    }
    return inner_->PersistAliveConnection(c);
}

int GraphDS::PersistRequest(const datastore::Request* r) {
    if (!r) return SG_EINVAL;
    sg_event ev; std::memset(&ev, 0, sizeof ev);
    bool tls = r->Tls;
    ev.protocol = ProtocolId(r->Protocol, &tls);
    // ReverseDirection() was applied for AMQP DELIVER / Redis PUSHED_EVENT (data.go:1110-1112,1151-1153): undo it,
    // K1 re-applies it after its own join
    const bool rev = (ev.protocol == SG_PROTO_AMQP && r->Method == "DELIVER") || (ev.protocol == SG_PROTO_REDIS && r->Method == "PUSHED_EVENT");
    const std::string& sip = rev ? r->ToIP : r->FromIP; const std::string& dip = rev ? r->FromIP : r->ToIP;
    const std::string& dtype = rev ? r->FromType : r->ToType; const std::string& duid = rev ? r->FromUID : r->ToUID;
    if (!ParseIPv4(sip, &ev.saddr) || !ParseIPv4(dip, &ev.daddr)) return SG_OK;   // not IPv4: nothing the join could do
    ev.status = (uint16_t)(r->StatusCode > 0xFFFFu ? 0xFFFFu : r->StatusCode);
    ev.flags = (tls ? SG_EV_TLS : 0) | (rev ? SG_EV_REVERSE : 0);
    ev.duration_ns = r->Latency;
    ev.write_time_ns = (uint64_t)r->StartTime * 1000000ull;       // already wall-clock ms; the engine clock is (0, 0) for this tap
    std::lock_guard<std::mutex> g(mu_);
    if (dtype == "outbound" && duid != dip) {                     // named by Host header (or reverse DNS): a label
        auto it = dto_labels_.find(duid);
        if (it == dto_labels_.end()) {
            std::vector<sg_event> tmp;                            // intern through the packer so both taps share one id space
            l7_req::L7Event fake; fake.ProtocolId = SG_PROTO_HTTP; fake.Daddr = ev.daddr;
            const std::string pl = "GET / HTTP/1.1\r\nHost: " + duid + "\r\n";
            fake.PayloadSize = (uint32_t)std::min(pl.size(), l7_req::kMaxPayload); std::memcpy(fake.Payload, pl.data(), fake.PayloadSize);
            if (!packer_.IsKnownIP(ev.daddr)) { packer_.Pack(fake, 1, &tmp); if (!tmp.empty()) it = dto_labels_.emplace(duid, tmp[0].host_label).first; }
        }
        if (it != dto_labels_.end()) ev.host_label = it->second;
    }
    return Append(ev);
}

int GraphDS::PersistKafkaEvent(const datastore::KafkaEvent* k) {
    if (!k) return SG_EINVAL;
    sg_event ev; std::memset(&ev, 0, sizeof ev);
    if (!ParseIPv4(k->FromIP, &ev.saddr) || !ParseIPv4(k->ToIP, &ev.daddr)) return SG_OK;
    ev.protocol = SG_PROTO_KAFKA; ev.status = 1;
    ev.flags = (k->Tls ? SG_EV_TLS : 0) | (k->Type == "CONSUME" ? SG_EV_CONSUME : 0);
    ev.duration_ns = k->Latency; ev.write_time_ns = (uint64_t)k->StartTime * 1000000ull;
    std::lock_guard<std::mutex> g(mu_);
    return Append(ev);
}

int GraphDS::IngestL7(const l7_req::L7Event& e, uint32_t kafka_msgs) {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<sg_event> tmp;
    packer_.Pack(e, kafka_msgs, &tmp);
    int rc = SG_OK;
    for (const sg_event& ev : tmp) { const int r2 = Append(ev); if (r2 != SG_OK) rc = r2; }
    return rc;
}

int GraphDS::IngestWire(const uint8_t* recs, size_t n, const uint32_t* kafka_msgs) {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<sg_event> tmp;
    int rc = SG_OK;
    for (size_t i = 0; i < n; i++) {
        tmp.clear();
        packer_.PackWire(recs + i * l7_req::kWireSize, kafka_msgs ? kafka_msgs[i] : 1u, &tmp);
        for (const sg_event& ev : tmp) { const int r2 = Append(ev); if (r2 != SG_OK) rc = r2; }
    }
    return rc;
}

long GraphDS::FlushWindow(int64_t window_end_ms) {
    std::vector<sg_edge_out> rows(max_edges_);
    std::vector<uint32_t> obips;
    size_t n = 0;
    std::vector<std::string> labels;
    {
        std::lock_guard<std::mutex> g(mu_);
        int rc = FlushBatchLocked();
        if (rc != SG_OK) return rc;
        api_.set_label_count(h_, (uint32_t)packer_.Labels().size());
        rc = api_.flush_window(h_, (uint64_t)window_end_ms, rows.data(), rows.size(), &n);
        if (rc != SG_OK) return rc;
        size_t no = 0;
        api_.window_outbound_ips(h_, nullptr, 0, &no);
        obips.resize(no);
        if (no) api_.window_outbound_ips(h_, obips.data(), no, &no);
        labels = packer_.Labels();
    }
    n = std::min(n, rows.size());
    std::vector<EdgeRow> out(n);
    auto name = [&](uint32_t ref, std::string* type, std::string* uid) {
        const uint32_t t = SG_REF_TYPE(ref), v = SG_REF_VALUE(ref);
        if (t == SG_REF_KNOWN && v < uid_of_.size()) { *type = kind_of_[v] == SG_NODE_SERVICE ? "service" : "pod"; *uid = uid_of_[v]; }
        else if (t == SG_REF_LABEL && v < labels.size()) { *type = "outbound"; *uid = labels[v]; }
        else if (t == SG_REF_OBIP && v < obips.size()) { *type = "outbound"; *uid = FormatIPv4(obips[v]); }
        else { *type = "unknown"; uid->clear(); }
    };
    {
        std::lock_guard<std::mutex> g(mu_);          // uid_of_ / kind_of_ are appended to by PersistPod / PersistService
        for (size_t i = 0; i < n; i++) {
            const sg_edge_out& r = rows[i]; EdgeRow& o = out[i];
            name(r.from_ref, &o.FromType, &o.FromUID); name(r.to_ref, &o.ToType, &o.ToUID);
            o.Count = r.count; o.ErrCount = r.err_count; o.SumNs = r.sum_ns; o.MaxNs = r.max_ns; o.SumSqUs = r.sumsq_us;
            o.Score = r.score; o.LatZ = r.lat_z; o.ErrRatio = r.err_ratio; o.Alive = r.alive;
        }
    }
    if (sink_) { const int rc = sink_->PersistEdges(window_end_ms, out); if (rc != 0) return rc; }
    return (long)n;
}

}  // namespace alaz
