// http2.hpp — host side of SURVEY.md §8 f-4 (first half): HTTP/2 events -> one packed event per request.
//
// The reference assembles a request from the HEADERS frames of the two directions of a stream
// (aggregator/data.go:544-810, processHttp2Frames) with one HPACK decoder per direction and connection
// (golang.org/x/net v0.20.0 http2/hpack, go.mod:118).  This is the same state machine for the drop-in:
//   hpack::Decoder     Decoder.Write semantics of that package (partial blocks carried over to the next
//                      Write, errors abandon the rest of a Write, size-update rule keyed on the
//                      connection's first field) — RFC 7541
//   Http2Assembler     frame-header walk (:618-627), first HEADERS frame of an event only (:741,:800),
//                      FrameArrival pairing by (pid, fd, stream) (:545-547), persistReq's local checks
//                      (:576-616), the minute sweep (:553-567), parser lifetime (processExit :362-377,
//                      processTcpConnect :484-494), the live-pid gate (processHttp2Event :1019-1033)
// The join (setFromToV2 with :authority as the host header) is K1 on the GPU, as for every protocol.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "l7_event.hpp"

namespace alaz {
namespace hpack {

struct HeaderField { std::string Name, Value; };

// HuffmanDecode appends to *out; false = ErrInvalidHuffman (EOS in the string, padding > 7 bits or not all ones)
bool HuffmanDecode(const uint8_t* p, size_t n, std::string* out);
// HuffmanEncode: for payload builders and tests
void HuffmanEncode(const std::string& s, std::string* out);

class Decoder {
public:
    // the views are valid during the call only (they point into the tables or into the decoder's scratch strings)
    using EmitFunc = std::function<void(std::string_view name, std::string_view value)>;
    explicit Decoder(uint32_t max_dynamic_table_size = 4096) : max_size_(max_dynamic_table_size), allowed_max_(max_dynamic_table_size) {}
    void SetEmitFunc(EmitFunc f) { emit_ = std::move(f); }
    // false on a decoding error (the caller in the reference ignores it; state stays usable)
    bool Write(const uint8_t* p, size_t n);
    size_t DynamicTableLen() const { return table_.size(); }
    uint32_t DynamicTableSize() const { return size_; }
    const HeaderField& DynamicTableAt(size_t i) const { return table_[i]; }     // 0 = newest

private:
    enum Result { kOk, kNeedMore, kError };
    Result ParseField(const uint8_t* p, size_t n, size_t* used);
    bool At(uint64_t i, std::string_view* name, std::string_view* value) const;
    void Add(HeaderField f);
    void Evict();

    std::deque<HeaderField> table_;      // front = newest
    uint32_t size_ = 0, max_size_, allowed_max_;
    std::string save_;                   // tail of a block that ended inside a field
    std::string name_buf_, value_buf_;   // decoded literals of the field being parsed
    bool first_field_ = true;
    EmitFunc emit_;
};

}  // namespace hpack

struct Http2Request {                    // what persistReq hands to setFromToV2 / PersistRequest
    std::string Method, Path, Authority;
    const char* Protocol = "HTTP2";      // "HTTP2" | "HTTPS" | "gRPC"
    uint32_t StatusCode = 0;
    uint64_t Latency = 0;
};

class Http2Assembler {
public:
    // One HTTP2 L7 event (MethodId 1 = CLIENT_FRAME, 2 = SERVER_FRAME).  true when this event completed a
    // request that passes persistReq's own checks; the request is in *out.
    bool OnEvent(const l7_req::L7Event& e, Http2Request* out);
    void ProcExec(uint32_t pid) { live_.insert(pid); }
    void ProcExit(uint32_t pid);
    void ConnClosed(uint32_t pid, uint64_t fd) { parsers_.erase(ConnKey{pid, fd}); }
    void Sweep();                        // the one-minute ticker: streams with one side only are forgotten

    size_t Pending() const { return frames_.size(); }
    size_t Parsers() const { return parsers_.size(); }
    uint64_t DroppedNotLive() const { return dropped_not_live_; }
    uint64_t DroppedUnparsed() const { return dropped_unparsed_; }
    uint64_t DroppedTime() const { return dropped_time_; }

private:
    struct ConnKey { uint32_t pid; uint64_t fd; bool operator==(const ConnKey& o) const { return pid == o.pid && fd == o.fd; } };
    struct StreamKey { uint32_t pid; uint64_t fd; uint32_t stream; bool operator==(const StreamKey& o) const { return pid == o.pid && fd == o.fd && stream == o.stream; } };
    struct KeyHash {
        size_t operator()(const ConnKey& k) const { return std::hash<uint64_t>()(k.fd * 0x9E3779B97F4A7C15ull ^ k.pid); }
        size_t operator()(const StreamKey& k) const { return std::hash<uint64_t>()((k.fd * 0x9E3779B97F4A7C15ull ^ k.pid) * 31 + k.stream); }
    };
    struct Parser { hpack::Decoder client{4096}, server{4096}; };
    struct FrameArrival {
        bool client = false, server = false, grpc = false;
        std::string method, path, authority;
        uint64_t client_write_ns = 0;            // req.Latency until persistReq
        uint32_t status = 0, grpc_status = 0;
    };
    bool Persist(const FrameArrival& f, const l7_req::L7Event& e, Http2Request* out);

    std::unordered_map<ConnKey, Parser, KeyHash> parsers_;
    std::unordered_map<StreamKey, FrameArrival, KeyHash> frames_;
    std::unordered_set<uint32_t> live_;
    uint64_t dropped_not_live_ = 0, dropped_unparsed_ = 0, dropped_time_ = 0;
};

uint32_t GoAtoiU32(std::string_view s);        // uint32(s) of `s, _ := strconv.Atoi(v)`

}  // namespace alaz
