// l7_event.hpp — the user-space L7 event of the reference (ebpf/l7_req/l7.go:396-417) and the
// kernel wire record it is built from (struct l7_event, ebpf/c/l7.c:19-47; bpfL7Event l7.go:345-369).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

namespace alaz {
namespace l7_req {

constexpr size_t kWireSize = 1096;
constexpr size_t kMaxPayload = 1024;

// BPF protocol enum — l7.go:19-29 (== SG_PROTO_* of servicegraph.h)
enum : uint8_t { BPF_L7_PROTOCOL_UNKNOWN = 0, BPF_L7_PROTOCOL_HTTP, BPF_L7_PROTOCOL_AMQP, BPF_L7_PROTOCOL_POSTGRES, BPF_L7_PROTOCOL_HTTP2,
                 BPF_L7_PROTOCOL_REDIS, BPF_L7_PROTOCOL_KAFKA, BPF_L7_PROTOCOL_MYSQL, BPF_L7_PROTOCOL_MONGO };

struct L7Event {
    uint64_t Fd = 0; uint32_t Pid = 0; uint32_t Status = 0; uint64_t Duration = 0;
    uint8_t ProtocolId = 0, MethodId = 0;   // the BPF enums; Protocol()/Method() give the reference's strings
    bool Tls = false;
    uint8_t Payload[kMaxPayload] = {0}; uint32_t PayloadSize = 0;
    bool PayloadReadComplete = false, Failed = false;
    uint64_t WriteTimeNs = 0;
    int16_t KafkaApiVersion = 0; uint32_t MySqlPrepStmtId = 0;
    uint32_t Saddr = 0; uint16_t Sport = 0; uint32_t Daddr = 0; uint16_t Dport = 0;

    const char* Protocol() const {   // L7ProtocolConversion.String(), l7.go:47-72
        static const char* t[] = {"UNKNOWN", "HTTP", "AMQP", "POSTGRES", "HTTP2", "REDIS", "KAFKA", "MYSQL", "MONGO"};
        return ProtocolId <= 8 ? t[ProtocolId] : "Unknown";
    }
    const char* Method() const {     // per-protocol conversions, l7.go:200-330
        static const char* http[] = {"Unknown", "GET", "POST", "PUT", "PATCH", "DELETE", "HEAD", "CONNECT", "OPTIONS", "TRACE"};
        const uint8_t m = MethodId;
        switch (ProtocolId) {
        case BPF_L7_PROTOCOL_HTTP: return (m >= 1 && m <= 9) ? http[m] : "Unknown";
        case BPF_L7_PROTOCOL_AMQP: return m == 1 ? "PUBLISH" : m == 2 ? "DELIVER" : "Unknown";
        case BPF_L7_PROTOCOL_POSTGRES: return m == 1 ? "CLOSE_OR_TERMINATE" : m == 2 ? "SIMPLE_QUERY" : m == 3 ? "EXTENDED_QUERY" : "Unknown";
        case BPF_L7_PROTOCOL_HTTP2: return m == 1 ? "CLIENT_FRAME" : m == 2 ? "SERVER_FRAME" : "Unknown";
        case BPF_L7_PROTOCOL_REDIS: return m == 1 ? "COMMAND" : m == 2 ? "PUSHED_EVENT" : m == 3 ? "PING" : "Unknown";
        case BPF_L7_PROTOCOL_KAFKA: return m == 1 ? "PRODUCE_REQUEST" : m == 2 ? "FETCH_RESPONSE" : "Unknown";
        case BPF_L7_PROTOCOL_MYSQL: return m == 1 ? "TEXT_QUERY" : m == 2 ? "PREPARE_STMT" : m == 3 ? "EXEC_STMT" : m == 4 ? "STMT_CLOSE" : "Unknown";
        default: return "Unknown";
        }
    }
};

// what L7Prog.Consume does with one perf sample (l7.go:704-762), minus the 1 KiB copy when only
// the header fields are wanted (copy_payload = false is the f-1 "packed producer" fast path).
inline void DecodeWire(const uint8_t* r, L7Event* e, bool copy_payload = true) {
    auto rd64 = [&](size_t o) { uint64_t v; std::memcpy(&v, r + o, 8); return v; };
    auto rd32 = [&](size_t o) { uint32_t v; std::memcpy(&v, r + o, 4); return v; };
    auto rd16 = [&](size_t o) { uint16_t v; std::memcpy(&v, r + o, 2); return v; };
    e->Fd = rd64(0); e->WriteTimeNs = rd64(8); e->Pid = rd32(16); e->Status = rd32(20); e->Duration = rd64(24);
    e->ProtocolId = r[32]; e->MethodId = r[33];
    e->PayloadSize = rd32(1060); if (e->PayloadSize > kMaxPayload) e->PayloadSize = kMaxPayload;
    if (copy_payload) std::memcpy(e->Payload, r + 36, kMaxPayload);
    e->PayloadReadComplete = r[1064] != 0; e->Failed = r[1065] != 0; e->Tls = r[1066] != 0;
    e->KafkaApiVersion = (int16_t)rd16(1068); e->MySqlPrepStmtId = rd32(1072);
    e->Saddr = rd32(1076); e->Sport = rd16(1080); e->Daddr = rd32(1084); e->Dport = rd16(1088);
}

}  // namespace l7_req
}  // namespace alaz
