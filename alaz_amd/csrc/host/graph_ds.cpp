#include "graph_ds.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>

namespace alaz {

bool SgApi::FromLibrary(void* dl, SgApi* o) {
    if (!dl || !o) return false;
#define SG_SYM(field, name) do { o->field = reinterpret_cast<decltype(o->field)>(dlsym(dl, name)); if (!o->field) return false; } while (0)
    SG_SYM(create, "sg_create"); SG_SYM(destroy, "sg_destroy"); SG_SYM(upsert_pod, "sg_upsert_pod"); SG_SYM(delete_pod, "sg_delete_pod");
    SG_SYM(upsert_service, "sg_upsert_service"); SG_SYM(delete_service, "sg_delete_service"); SG_SYM(set_label_count, "sg_set_label_count");
    SG_SYM(ingest, "sg_ingest"); SG_SYM(flush_window, "sg_flush_window"); SG_SYM(window_outbound_ips, "sg_window_outbound_ips");
    o->flush_window_view = reinterpret_cast<decltype(o->flush_window_view)>(dlsym(dl, "sg_flush_window_view"));   // optional
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

GraphDS::GraphDS(datastore::DataStore* inner, const SgApi& api, sg_handle h, EdgeSink* sink, size_t max_edges, size_t batch,
                 uint32_t max_known_nodes, bool divert_requests)
    : inner_(inner), api_(api), h_(h), sink_(sink), max_edges_(max_edges), batch_cap_(batch), max_known_(max_known_nodes), divert_(divert_requests) {
    for (Shard& s : shards_) s.batch.reserve(batch);
}
GraphDS::~GraphDS() = default;

uint32_t GraphDS::Intern(const std::string& uid, uint8_t kind) {
    auto it = ids_.find(uid);
    if (it != ids_.end()) { kind_of_[it->second] = kind; return it->second; }
    uint32_t id;
    if (!free_ids_.empty()) { id = free_ids_.back(); free_ids_.pop_back(); uid_of_[id] = uid; kind_of_[id] = kind; refs_[id] = 0; }
    else {
        if (uid_of_.size() >= max_known_) return kNoId;            // the engine's id space is full (sg_config.max_known_nodes)
        id = (uint32_t)uid_of_.size();
        uid_of_.push_back(uid); kind_of_.push_back(kind); refs_.push_back(0);
    }
    ids_.emplace(uid, id);
    live_ids_++;
    return id;
}
void GraphDS::UnbindIP(std::unordered_map<uint32_t, uint32_t>& m, uint32_t ip) {
    auto it = m.find(ip);
    if (it == m.end()) return;
    const uint32_t id = it->second;
    m.erase(it);
    if (--refs_[id] == 0) retired_.push_back(id);                  // no IP names it any more; the open window may still do
}
void GraphDS::BindIP(std::unordered_map<uint32_t, uint32_t>& m, uint32_t ip, uint32_t id) {
    auto it = m.find(ip);
    if (it != m.end() && it->second == id) return;
    refs_[id]++;                                                     // before the unbind: re-binding an id's only IP must not retire it
    if (it != m.end()) { const uint32_t old = it->second; it->second = id; if (--refs_[old] == 0) retired_.push_back(old); }
    else m.emplace(ip, id);
}

// processPod keeps PodIPToPodUid (aggregator/persist.go:55-71); pods without an IP never reach the
// datastore (persist.go:37-40), an empty IP here is ignored for the same reason.
int GraphDS::PersistPod(const datastore::Pod& pod, const std::string& et) {
    uint32_t ip; int erc = SG_OK;
    if (!pod.IP.empty() && ParseIPv4(pod.IP, &ip)) {
        const bool up = et == datastore::ADD || et == datastore::UPDATE;
        if (up || et == datastore::DELETE_) {
            std::lock_guard<std::mutex> g(id_mu_);
            if (up) {
                const uint32_t id = Intern(pod.UID, SG_NODE_POD);
                erc = id == kNoId ? SG_ENOSPC : api_.upsert_pod(h_, ip, id);
                if (erc == SG_OK) BindIP(pod_ip_id_, ip, id);
            } else { api_.delete_pod(h_, ip); UnbindIP(pod_ip_id_, ip); }     // (a DELETE never creates an id)
        }
        if (erc == SG_OK && (up || et == datastore::DELETE_)) { std::lock_guard<std::mutex> g(pk_mu_); packer_.SetPodIP(ip, up); }
        if (erc != SG_OK) engine_errors_++;
    }
    const int rc = inner_->PersistPod(pod, et);
    return rc != 0 ? rc : erc;
}

// processSvc keys ServiceIPToServiceUid on Spec.ClusterIP (persist.go:114-130); the DTO carries it as
// ClusterIPs[0] (ClusterIP itself is never filled in, persist.go:105-112).
int GraphDS::PersistService(const datastore::Service& svc, const std::string& et) {
    const std::string& ips = !svc.ClusterIPs.empty() ? svc.ClusterIPs[0] : svc.ClusterIP;
    uint32_t ip; int erc = SG_OK;
    if (ParseIPv4(ips, &ip)) {
        const bool up = et == datastore::ADD || et == datastore::UPDATE;
        if (up || et == datastore::DELETE_) {
            std::lock_guard<std::mutex> g(id_mu_);
            if (up) {
                const uint32_t id = Intern(svc.UID, SG_NODE_SERVICE);
                erc = id == kNoId ? SG_ENOSPC : api_.upsert_service(h_, ip, id);
                if (erc == SG_OK) BindIP(svc_ip_id_, ip, id);
            } else { api_.delete_service(h_, ip); UnbindIP(svc_ip_id_, ip); }
        }
        if (erc == SG_OK && (up || et == datastore::DELETE_)) { std::lock_guard<std::mutex> g(pk_mu_); packer_.SetServiceIP(ip, up); }
        if (erc != SG_OK) engine_errors_++;
    }
    const int rc = inner_->PersistService(svc, et);
    return rc != 0 ? rc : erc;
}

int GraphDS::FlushShard(Shard& s) {
    if (s.batch.empty()) return SG_OK;
    const int rc = api_.ingest(h_, s.batch.data(), s.batch.size());   // copies; the engine has its own lock
    if (rc == SG_EAGAIN) batches_dropped_++;          // never block the aggregator (the reference would: backend.go:844)
    s.batch.clear();
    return rc == SG_EAGAIN ? SG_OK : rc;
}

// events of one caller go to the shard its thread hashes to: goroutines on different OS threads do not meet here
int GraphDS::Append(const sg_event* ev, size_t n) {
    if (n == 0) return SG_OK;
    Shard& s = shards_[std::hash<std::thread::id>()(std::this_thread::get_id()) % kShards];
    std::lock_guard<std::mutex> g(s.mu);
    int rc = SG_OK;
    for (size_t i = 0; i < n; i++) {
        s.batch.push_back(ev[i]);
        if (s.batch.size() >= batch_cap_) { const int r2 = FlushShard(s); if (r2 != SG_OK) rc = r2; }
    }
    offered_ += n;
    return rc;
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
        Append(&ev, 1);
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
    if (dtype == "outbound" && duid != dip) {                     // named by Host header (or reverse DNS): a label
        std::lock_guard<std::mutex> g(pk_mu_);
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
    const int rc = Append(&ev, 1);
    if (!divert_) { const int r2 = inner_->PersistRequest(r); if (r2 != 0) return r2; }   // additive by default
    return rc;
}

int GraphDS::PersistKafkaEvent(const datastore::KafkaEvent* k) {
    if (!k) return SG_EINVAL;
    sg_event ev; std::memset(&ev, 0, sizeof ev);
    if (!ParseIPv4(k->FromIP, &ev.saddr) || !ParseIPv4(k->ToIP, &ev.daddr)) return SG_OK;
    ev.protocol = SG_PROTO_KAFKA; ev.status = 1;
    ev.flags = (k->Tls ? SG_EV_TLS : 0) | (k->Type == "CONSUME" ? SG_EV_CONSUME : 0);
    ev.duration_ns = k->Latency; ev.write_time_ns = (uint64_t)k->StartTime * 1000000ull;
    const int rc = Append(&ev, 1);
    if (!divert_) { const int r2 = inner_->PersistKafkaEvent(k); if (r2 != 0) return r2; }
    return rc;
}

// The payload-dependent part (Host header, SQL filters, HPACK, Kafka decode) runs under the packer's lock only; the packed
// events are appended to the caller's shard afterwards, outside it.
int GraphDS::IngestL7(const l7_req::L7Event& e, uint32_t kafka_msgs) {
    std::vector<sg_event> tmp;
    { std::lock_guard<std::mutex> g(pk_mu_); packer_.Pack(e, kafka_msgs, &tmp); }
    return Append(tmp.data(), tmp.size());
}

int GraphDS::IngestWire(const uint8_t* recs, size_t n, const uint32_t* kafka_msgs) {
    std::vector<sg_event> tmp;
    int rc = SG_OK;
    constexpr size_t kChunk = 256;                                // records packed per hold of the packer lock
    for (size_t i0 = 0; i0 < n; i0 += kChunk) {
        tmp.clear();
        {
            std::lock_guard<std::mutex> g(pk_mu_);
            for (size_t i = i0; i < n && i < i0 + kChunk; i++) packer_.PackWire(recs + i * l7_req::kWireSize, kafka_msgs ? kafka_msgs[i] : 1u, &tmp);
        }
        const int r2 = Append(tmp.data(), tmp.size());
        if (r2 != SG_OK) rc = r2;
    }
    return rc;
}

long GraphDS::FlushWindow(int64_t window_end_ms) {
    std::lock_guard<std::mutex> fg(flush_mu_);
    std::vector<sg_edge_out> own;                  // only without sg_flush_window_view: a pageable copy of up to max_edges rows
    const sg_edge_out* rows = nullptr;
    std::vector<uint32_t> obips;
    size_t n = 0;
    std::vector<std::string> labels;
    std::vector<uint32_t> retire_now;
    // no GraphDS lock is held across the engine's window pipeline (sg_flush_window synchronises with the device): the
    // Persist* / Ingest* callers keep appending to their shards, which go to the NEXT window
    for (Shard& s : shards_) { std::lock_guard<std::mutex> g(s.mu); const int rc = FlushShard(s); if (rc != SG_OK) return rc; }
    { std::lock_guard<std::mutex> g(pk_mu_); labels = packer_.Labels(); }
    { std::lock_guard<std::mutex> g(id_mu_); retire_now.swap(retired_); }     // ids whose last IP went away before this window closed
    api_.set_label_count(h_, (uint32_t)labels.size());
    int rc;
    if (api_.flush_window_view) rc = api_.flush_window_view(h_, (uint64_t)window_end_ms, &rows, &n);     // rows stay valid: flush_mu_ is held
    else { own.resize(max_edges_); rc = api_.flush_window(h_, (uint64_t)window_end_ms, own.data(), own.size(), &n); rows = own.data(); n = std::min(n, own.size()); }
    if (rc != SG_OK) return rc;
    // the label table again, AFTER the window has closed: a feeder may have interned a label and pushed its event into this
    // window between the first snapshot and the close (labels are append-only, so the later snapshot is a superset)
    { std::lock_guard<std::mutex> g(pk_mu_); if (packer_.Labels().size() != labels.size()) labels = packer_.Labels(); }
    {
        size_t no = 0;
        api_.window_outbound_ips(h_, nullptr, 0, &no);
        obips.resize(no);
        if (no) api_.window_outbound_ips(h_, obips.data(), no, &no);
    }
    std::vector<EdgeRow> out(n);
    auto name = [&](uint32_t ref, std::string* type, std::string* uid) {
        const uint32_t t = SG_REF_TYPE(ref), v = SG_REF_VALUE(ref);
        if (t == SG_REF_KNOWN && v < uid_of_.size()) { *type = kind_of_[v] == SG_NODE_SERVICE ? "service" : "pod"; *uid = uid_of_[v]; }
        else if (t == SG_REF_LABEL && v < labels.size()) { *type = "outbound"; *uid = labels[v]; }
        else if (t == SG_REF_OBIP && v < obips.size()) { *type = "outbound"; *uid = FormatIPv4(obips[v]); }
        else { *type = "unknown"; uid->clear(); }
    };
    {
        std::lock_guard<std::mutex> g(id_mu_);       // uid_of_ / kind_of_ are appended to by PersistPod / PersistService
        for (size_t i = 0; i < n; i++) {
            const sg_edge_out& r = rows[i]; EdgeRow& o = out[i];
            name(r.from_ref, &o.FromType, &o.FromUID); name(r.to_ref, &o.ToType, &o.ToUID);
            o.Count = r.count; o.ErrCount = r.err_count; o.SumNs = r.sum_ns; o.MaxNs = r.max_ns; o.SumSqUs = r.sumsq_us;
            o.Score = r.score; o.LatZ = r.lat_z; o.ErrRatio = r.err_ratio; o.Alive = r.alive; o.P50Us = r.p50_us; o.P99Us = r.p99_us;
        }
    }
    {   // the window that could still name the retired ids has been read: they may be handed out again, unless an IP
        // was bound to them in the meantime
        std::lock_guard<std::mutex> g(id_mu_);
        for (uint32_t id : retire_now) {
            if (refs_[id] != 0 || uid_of_[id].empty()) continue;
            auto it = ids_.find(uid_of_[id]);
            if (it != ids_.end() && it->second == id) ids_.erase(it);
            uid_of_[id].clear(); kind_of_[id] = 0; free_ids_.push_back(id); live_ids_--;
        }
    }
    if (sink_) { const int rc2 = sink_->PersistEdges(window_end_ms, out); if (rc2 != 0) return rc2; }
    return (long)n;
}

}  // namespace alaz
