// graph_ds.hpp — GraphDS: a datastore.DataStore decorator that feeds the MI355X ServiceGraph engine.
//
// It is what INTEGRATION.md's Go `GraphDS` is, written in C++ because no Go toolchain exists in the
// build environment: it wraps an inner DataStore (the reference's BackendDS), forwards the eight k8s
// resource calls unchanged, mirrors pod / service IP changes into the engine's join tables
// (sg_upsert_* / sg_delete_*, the analogue of aggregator/persist.go:55-71,114-130), turns the per-request
// calls into packed events (sg_ingest) and, once per window, reads back one scored row per edge
// (sg_flush_window) and hands them to an EdgeSink.
//
// Two taps exist, as in SURVEY.md §8b:
//   * PersistRequest / PersistKafkaEvent — the datastore boundary itself (the reference aggregator has
//     already run setFromToV2; the engine repeats the join on the GPU from FromIP / ToIP);
//   * IngestL7 — one step earlier, straight from l7_req.L7Event, so that the reference's CPU join is
//     not needed at all (the L7Packer does the payload-dependent part on the host).
#pragma once
#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/servicegraph.h"
#include "datastore.hpp"
#include "l7_event.hpp"
#include "packer.hpp"

namespace alaz {

// the C ABI as a table of function pointers: the real library (dlsym) in production, a recording
// stand-in in the host-logic tests.
struct SgApi {
    int (*create)(const sg_config*, sg_handle*) = nullptr;
    int (*destroy)(sg_handle) = nullptr;
    int (*upsert_pod)(sg_handle, uint32_t, uint32_t) = nullptr;
    int (*delete_pod)(sg_handle, uint32_t) = nullptr;
    int (*upsert_service)(sg_handle, uint32_t, uint32_t) = nullptr;
    int (*delete_service)(sg_handle, uint32_t) = nullptr;
    int (*set_label_count)(sg_handle, uint32_t) = nullptr;
    int (*ingest)(sg_handle, const sg_event*, size_t) = nullptr;
    int (*flush_window)(sg_handle, uint64_t, sg_edge_out*, size_t, size_t*) = nullptr;
    int (*flush_window_view)(sg_handle, uint64_t, const sg_edge_out**, size_t*) = nullptr;   // optional (absent in a recording stand-in): rows stay in the engine's pinned buffer
    int (*window_outbound_ips)(sg_handle, uint32_t*, size_t, size_t*) = nullptr;
    const char* (*last_error)(sg_handle) = nullptr;
    static bool FromLibrary(void* dl_handle, SgApi* out);      // dlsym of every entry; false if one is missing
};

struct EdgeRow {                       // one edge of a closed window, in the reference's vocabulary
    std::string FromType, FromUID, ToType, ToUID;
    uint32_t Count = 0, ErrCount = 0;
    uint64_t SumNs = 0, MaxNs = 0, SumSqUs = 0;
    float Score = 0, LatZ = 0, ErrRatio = 0;
    uint32_t Alive = 0;                // open connections reported on the edge in the window
    uint32_t P50Us = 0, P99Us = 0;     // latency percentiles off the edge's log2 histogram (0 unless SG_CFG_EDGE_HISTOGRAM)
};

class EdgeSink {
public:
    virtual ~EdgeSink() = default;
    virtual int PersistEdges(int64_t window_end_ms, const std::vector<EdgeRow>& rows) = 0;
};

bool ParseIPv4(const std::string& s, uint32_t* out);     // "a.b.c.d" -> a<<24|b<<16|c<<8|d
std::string FormatIPv4(uint32_t ip);                      // IntToIPv4().String(), aggregator/data.go:1751-1758

class GraphDS : public datastore::DataStore {
public:
    // max_known_nodes = sg_config.max_known_nodes (the id space the engine was created with).  divert_requests: false
    // (default) = additive — PersistRequest / PersistKafkaEvent also reach the inner data store, so a backend without an
    // "/edges/" route keeps receiving its per-request rows; true = the engine's per-edge rows replace them.
    GraphDS(datastore::DataStore* inner, const SgApi& api, sg_handle h, EdgeSink* sink, size_t max_edges, size_t batch = 4096,
            uint32_t max_known_nodes = 0x3FFFFFFFu, bool divert_requests = false);
    ~GraphDS() override;

    int PersistPod(const datastore::Pod& pod, const std::string& eventType) override;
    int PersistService(const datastore::Service& service, const std::string& eventType) override;
    int PersistReplicaSet(const datastore::ReplicaSet& rs, const std::string& et) override { return inner_->PersistReplicaSet(rs, et); }
    int PersistDeployment(const datastore::Deployment& d, const std::string& et) override { return inner_->PersistDeployment(d, et); }
    int PersistEndpoints(const datastore::Endpoints& e, const std::string& et) override { return inner_->PersistEndpoints(e, et); }
    int PersistContainer(const datastore::Container& c, const std::string& et) override { return inner_->PersistContainer(c, et); }
    int PersistDaemonSet(const datastore::DaemonSet& ds, const std::string& et) override { return inner_->PersistDaemonSet(ds, et); }
    int PersistStatefulSet(const datastore::StatefulSet& ss, const std::string& et) override { return inner_->PersistStatefulSet(ss, et); }
    int PersistRequest(const datastore::Request* request) override;
    int PersistKafkaEvent(const datastore::KafkaEvent* request) override;
    // an open TCP connection (sendOpenConnection, data.go:1628-1679): forwarded, and fed to the engine as an
    // SG_EV_ALIVE record — the join (source must be a pod, service before pod, else the IP) runs on the GPU
    int PersistAliveConnection(const datastore::AliveConnection* conn) override;

    // earlier tap: the raw L7 event (what processL7 receives, aggregator/data.go:1364-1383)
    int IngestL7(const l7_req::L7Event& e, uint32_t kafka_msgs = 1);
    // n perf records of l7_req::kWireSize bytes each, through L7Packer::PackWire (f-1: no 1 KiB copy per event)
    int IngestWire(const uint8_t* recs, size_t n, const uint32_t* kafka_msgs = nullptr);

    // process / connection lifecycle as far as the payload parsers need it (aggregator/data.go:354-401, :484-503,
    // :553-567): only events of live pids are assembled; a closed connection or an exited process drops its HPACK state
    // and its remembered Postgres statements
    void ProcExec(uint32_t pid) { std::lock_guard<std::mutex> g(pk_mu_); packer_.Http2().ProcExec(pid); }
    void ProcExit(uint32_t pid) { std::lock_guard<std::mutex> g(pk_mu_); packer_.ProcExit(pid); }
    void ConnClosed(uint32_t pid, uint64_t fd) { std::lock_guard<std::mutex> g(pk_mu_); packer_.ConnClosed(pid, fd); }
    void SetKafkaDecode(bool on) { std::lock_guard<std::mutex> g(pk_mu_); packer_.SetKafkaDecode(on); }
    void SweepHttp2() { std::lock_guard<std::mutex> g(pk_mu_); packer_.Http2().Sweep(); }

    // close the window: pending batches -> engine, K2..K5, rows -> sink.  Returns the number of edges or < 0.
    long FlushWindow(int64_t window_end_ms);

    uint64_t EventsOffered() const { return offered_.load(); }
    uint64_t BatchesDropped() const { return batches_dropped_.load(); }
    uint64_t EngineErrors() const { return engine_errors_.load(); }   // sg_upsert_* failures (SG_ENOSPC: id space / join table full)
    size_t LiveIds() const { return live_ids_; }
    const std::vector<std::string>& Labels() const { return packer_.Labels(); }
    const L7Packer& Packer() const { return packer_; }

private:
    static constexpr uint32_t kNoId = 0xFFFFFFFFu;
    static constexpr int kShards = 8;
    struct Shard { std::mutex mu; std::vector<sg_event> batch; };

    // node ids (under id_mu_): one per UID that currently owns at least one IP in the join tables.  An id whose last IP
    // is gone is retired, and handed out again once the window that may still name it has been flushed.
    uint32_t Intern(const std::string& uid, uint8_t kind);            // kNoId when the id space is exhausted
    void BindIP(std::unordered_map<uint32_t, uint32_t>& m, uint32_t ip, uint32_t id);   // m[ip] = id, reference counts follow
    void UnbindIP(std::unordered_map<uint32_t, uint32_t>& m, uint32_t ip);
    int Append(const sg_event* ev, size_t n);
    int FlushShard(Shard& s);                          // s.mu held

    datastore::DataStore* inner_;
    SgApi api_; sg_handle h_; EdgeSink* sink_;
    size_t max_edges_, batch_cap_;
    uint32_t max_known_; bool divert_;
    Shard shards_[kShards];                            // event batches, one per feeder-thread hash: appends do not serialise
    std::mutex id_mu_;                                 // ids_, uid_of_, kind_of_, refs_, the IP mirrors
    std::unordered_map<std::string, uint32_t> ids_;    // UID -> node id
    std::vector<std::string> uid_of_; std::vector<uint8_t> kind_of_; std::vector<uint32_t> refs_;
    std::unordered_map<uint32_t, uint32_t> pod_ip_id_, svc_ip_id_;   // ip -> id, what the engine's join tables hold
    std::vector<uint32_t> free_ids_, retired_;         // retired_: no IP left, reusable after the next FlushWindow
    size_t live_ids_ = 0;
    std::mutex pk_mu_;                                 // packer_ (labels, prepared statements, HPACK state), dto_labels_
    L7Packer packer_;
    std::unordered_map<std::string, uint32_t> dto_labels_;   // labels seen through the PersistRequest tap share the packer's id space
    std::mutex flush_mu_;                              // one FlushWindow at a time
    std::atomic<uint64_t> offered_{0}, batches_dropped_{0}, engine_errors_{0};
};

}  // namespace alaz
