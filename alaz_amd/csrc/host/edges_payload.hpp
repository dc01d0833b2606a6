// edges_payload.hpp — SURVEY.md §8 f-3: the egress format of aggregated edges.
//
// The reference ships one ReqInfo row per request ([16]interface{} -> a positional JSON array, datastore/payload.go:109-134)
// in batches wrapped with Metadata (payload.go:3-8, built in backend.go:454-464) and POSTs them to "/requests/"
// (backend.go:180, 591-632).  An edge row per window replaces O(requests) rows by O(edges) rows; the payload keeps the
// reference's conventions so that a backend can take it beside RequestsPayload:
//
//   {"metadata":{"monitoring_id":..,"idempotency_key":..,"node_id":..,"alaz_version":..},
//    "window_end":<ms>,"edges":[[...13 slots...],...]}                      endpoint "/edges/"
//
//   slot 0 FromType  1 FromUID  2 ToType  3 ToUID  4 Count  5 ErrCount  6 SumLatencyNs  7 MaxLatencyNs  8 SumSqLatencyUs
//        9 AliveConnections  10 Score  11 LatencyZ  12 ErrRatio
//
// Strings are escaped the way encoding/json does by default (HTML-safe: < > & as < > &; U+2028/9; invalid
// UTF-8 -> U+FFFD); floats are written in their shortest round-trip form.  Sending the bytes (HTTP client, retries) is the
// inner data store's business and stays out of scope.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "graph_ds.hpp"

namespace alaz {

struct PayloadMetadata { std::string MonitoringID, IdempotencyKey, NodeID, AlazVersion; };   // datastore/payload.go:3-8

constexpr const char* kEdgesEndpoint = "/edges/";

void AppendJsonString(const std::string& s, std::string* out);                               // with the quotes
void AppendJsonFloat(float v, std::string* out);                                             // NaN / Inf -> null
// rows [first, first + n) of one window as one payload
std::string EdgesPayloadJson(const PayloadMetadata& md, int64_t window_end_ms, const EdgeRow* rows, size_t n);

// An EdgeSink that cuts a window into payloads of at most `batch` rows (the reference's batchSize idea, backend.go:591)
// and hands each to `post(endpoint, body)`; the idempotency key gets "-<k>" appended per payload.
class JsonEdgeSink : public EdgeSink {
public:
    using Post = std::function<int(const char* endpoint, const std::string& body)>;
    JsonEdgeSink(PayloadMetadata md, size_t batch, Post post) : md_(std::move(md)), batch_(batch ? batch : 1000), post_(std::move(post)) {}
    int PersistEdges(int64_t window_end_ms, const std::vector<EdgeRow>& rows) override;
    uint64_t PayloadsSent() const { return sent_; }
private:
    PayloadMetadata md_; size_t batch_; Post post_; uint64_t sent_ = 0;
};

}  // namespace alaz
