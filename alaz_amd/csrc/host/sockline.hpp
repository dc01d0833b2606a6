// sockline.hpp — TCP connect events -> per-(pid, fd) socket lines -> alive connections.
//
// Host-side mirror of the reference's connection bookkeeping, written from its behaviour:
//   SockInfo / SocketMap               aggregator/socket.go:10-41
//   TimestampedSocket / SocketLine     aggregator/sock_num_line.go:23-36, AddValue :62-81, GetValue :83-157,
//                                      DeleteUnused :159-208, sorted insert :311-322
//   TcpConnectEvent / BpfTcpEvent      ebpf/tcp_state/tcp.go:63-85, 226-243
//   processTcpConnect                  aggregator/data.go:404-506
//   sendOpenConnection / sweep         aggregator/data.go:1628-1716
// Addresses are numeric IPv4 here (a<<24|b<<16|c<<8|d, like L7Event.Saddr); the reference keeps dotted
// strings, the oracle (oracle/sockline.c) does too, and tests/test_sockline.py compares the two.
// Differences by design: the reference creates a process' socket map and an fd's socket line
// asynchronously (and may seed the line from /proc/<pid>/net/tcp) and re-queues the event meanwhile;
// here the line is created on demand and starts empty.  time.Now() is a parameter.
#pragma once
#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "datastore.hpp"

namespace alaz {

struct SockInfo {
    uint32_t Pid = 0; uint64_t Fd = 0;
    uint32_t Saddr = 0; uint16_t Sport = 0;
    uint32_t Daddr = 0; uint16_t Dport = 0;
};

struct TimestampedSocket {
    uint64_t Timestamp = 0;      // kernel time of the connect / close
    uint64_t LastMatch = 0;      // last time a request was matched to it (user time, ns)
    bool Open = false;           // false = the "nil SockInfo" close marker
    SockInfo Info;
};

enum class SockErr { Ok = 0, Empty = 1, ClosedLast = 2, NoSmaller = 3, Closed = 4 };

class SocketLine {
public:
    SocketLine(uint32_t pid, uint64_t fd) : pid_(pid), fd_(fd) {}
    void AddValue(uint64_t timestamp, const SockInfo* info);             // nullptr = close
    SockErr GetValue(uint64_t timestamp, uint64_t now_ns, SockInfo* out);
    void DeleteUnused();
    bool LastOpen(SockInfo* out) const;                                  // what sendOpenConnection looks at
    size_t Size() const;
    TimestampedSocket At(size_t i) const;
private:
    mutable std::mutex mu_;
    uint32_t pid_; uint64_t fd_;
    std::vector<TimestampedSocket> values_;
};

namespace tcp_state {
enum : uint32_t { kEstablished = 1, kConnectFailed = 2, kListen = 3, kListenClosed = 4, kClosed = 5 };   // tcp.go:19-25
constexpr size_t kWireSize = 64;
struct TcpConnectEvent {                                                   // tcp.go:75-85 (addresses numeric)
    uint64_t Fd = 0, Timestamp = 0; uint32_t Type = 0, Pid = 0;
    uint16_t SPort = 0, DPort = 0; uint32_t SAddr = 0, DAddr = 0;
};
TcpConnectEvent DecodeWire(const uint8_t* rec);                            // BpfTcpEvent, tcp.go:63-72
}  // namespace tcp_state

// clusterInfo.SocketMaps + the two aggregator routines that touch them
class ConnTracker {
public:
    // processTcpConnect; returns true if a value was added to a line
    bool ProcessTcpConnect(const tcp_state::TcpConnectEvent& e);
    // one tick of clearSocketLines: every line whose last value is an open socket is reported through
    // ds->PersistAliveConnection (resolution of UIDs is left to the data store — GraphDS does it on the
    // GPU from the IPs; FromType/ToType/UIDs stay empty here), then DeleteUnused.  Returns lines reported.
    size_t Sweep(int64_t now_ms, bool send_alive, datastore::DataStore* ds);
    SocketLine* Line(uint32_t pid, uint64_t fd);
    size_t Lines() const;
private:
    mutable std::mutex mu_;
    std::unordered_map<uint32_t, std::unordered_map<uint64_t, SocketLine*>> maps_;   // pid -> fd -> line
    std::vector<SocketLine*> all_;
public:
    ~ConnTracker();
};

}  // namespace alaz
