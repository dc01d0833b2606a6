// packer.hpp — host side of the hot path: one l7_req.L7Event -> packed sg_event(s).
//
// Everything that needs the 1 KiB payload stays here, exactly where the reference does it:
//   processHttpEvent   aggregator/data.go:1208-1249  (parseHttpPayload :508-531 -> Host header)
//   processPostgres..  :1323-1362 (parsePostgresCommand :1474-1556: drop on parse error; statement cache)
//   processMySQLEvent  :1287-1321 (parseMySQLCommand :1431-1472)
//   processMongoEvent  :1251-1285 (parseMongoEvent :1561-1617; recover() => keep the event)
//   processRedisEvent  :1120-1160, processAmqpEvent :1081-1118 (ReverseDirection for PUSHED_EVENT / DELIVER)
//   processKafkaEvent  :1035-1079 (one event per decoded message: decoded here with SetKafkaDecode(true), kafka.hpp,
//                      or counted by the caller)
//   processHttp2Event  :1019-1033 -> processHttp2Frames :544-810 (Http2Assembler, http2.hpp): an event is
//                      emitted when the second HEADERS frame of a stream arrives; latency = the distance of
//                      the two write times, status = :status or grpc-status, host label = :authority
// The join (setFromToV2) is NOT done here: it is K1 on the GPU.  The packer only decides whether
// an event reaches the join at all and which outbound label it would carry.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../../include/servicegraph.h"
#include "http2.hpp"
#include "kafka.hpp"
#include "l7_event.hpp"

namespace alaz {

// parseHttpPayload — aggregator/data.go:508-531 (strings.Split semantics)
void ParseHttpPayload(const char* req, size_t len, std::string* method, std::string* path, std::string* version, std::string* host);
// containsSQLKeywords — data.go:1624-1626 over the keyword list of :123
bool ContainsSQLKeywords(const uint8_t* s, size_t n);

class L7Packer {
public:
    // IPs currently present in the join tables: the Host header is interned only when the destination is in neither, i.e.
    // exactly when setFromToV2 would use it (:851-854).  Two plain sets with the reference's map semantics (persist.go:55-71,
    // 114-130: ADD and UPDATE store, DELETE erases) — not a count: ADD + UPDATE + DELETE of one pod leaves nothing behind.
    void SetPodIP(uint32_t ip, bool present) { if (present) pod_ips_.insert(ip); else pod_ips_.erase(ip); }
    void SetServiceIP(uint32_t ip, bool present) { if (present) svc_ips_.insert(ip); else svc_ips_.erase(ip); }
    bool IsKnownIP(uint32_t ip) const { return pod_ips_.count(ip) != 0 || svc_ips_.count(ip) != 0; }

    // Appends 0..n packed events for `e` to `out`.  kafka_msgs = number of messages a Kafka decoder on the
    // caller's side produced for this event (ignored for other protocols, and ignored when SetKafkaDecode(true):
    // then the payload is decoded here, kafka.hpp).  Returns the number appended.
    size_t Pack(const l7_req::L7Event& e, uint32_t kafka_msgs, std::vector<sg_event>* out);
    // The same straight from a 1096-byte perf record (the "packed producer"): the header fields are
    // read in place and the payload is copied — PayloadSize bytes of it, not the 1 KiB slot — only when the
    // protocol handler will look at it (HTTP towards an unknown address, the SQL/Mongo filters, HTTP/2, Kafka decode).
    size_t PackWire(const uint8_t* rec, uint32_t kafka_msgs, std::vector<sg_event>* out);

    const std::vector<std::string>& Labels() const { return labels_; }
    uint64_t DroppedParse() const { return dropped_parse_; }
    // a closed connection (processTcpConnect, data.go:484-503): its HPACK state goes, and every remembered Postgres
    // statement whose "pid-fd-name" key STARTS WITH "pid-fd" (the reference's HasPrefix: fd 7 also clears fd 70..79)
    void ConnClosed(uint32_t pid, uint64_t fd);
    // an exited process (processExit, data.go:363-401): its HTTP/2 parsers, and every remembered Postgres statement whose key
    // starts with the decimal pid (HasPrefix again: pid 12 also clears pid 120..129).  The reference's loop over mySqlStmts
    // iterates pgStmts' keys after they were deleted, so MySQL statements survive an exit there — and here.
    void ProcExit(uint32_t pid);
    size_t PgStatements() const { return pg_stmts_.size(); }
    void SetKafkaDecode(bool on) { kafka_decode_ = on; }
    Http2Assembler& Http2() { return h2_; }
    const Http2Assembler& Http2() const { return h2_; }

private:
    uint32_t InternLabel(const std::string& host);
    int ParsePostgres(const l7_req::L7Event& e, std::string* out);
    int ParseMySQL(const l7_req::L7Event& e, std::string* out);
    static int ParseMongo(const l7_req::L7Event& e, std::string* out);

    std::unordered_set<uint32_t> pod_ips_, svc_ips_;
    std::unordered_map<std::string, uint32_t> label_ids_;
    std::vector<std::string> labels_;
    std::unordered_map<std::string, std::string> pg_stmts_, mysql_stmts_;   // data.go pgStmts / mySqlStmts
    Http2Assembler h2_;
    bool kafka_decode_ = false;
    l7_req::L7Event scratch_;
    uint64_t dropped_parse_ = 0;
};

}  // namespace alaz
