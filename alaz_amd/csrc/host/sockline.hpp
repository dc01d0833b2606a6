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
//   NewSocketLine(fetch) / getConnectionInfo   aggregator/sock_num_line.go:38-54, 351-429 (SeedFromProc)
//   clearProc / processExit            aggregator/cluster.go:97-110, data.go:363-398 (ClearProc)
// Differences by design: the reference creates a process' socket map and an fd's socket line
// asynchronously and re-queues the event meanwhile; here the line is created on demand — seeded from
// <proc root>/<pid>/fd/<fd> and <proc root>/<pid>/net/tcp first when the tracker was given a proc root, as
// NewSocketLine(fetch = true) does — and the event applied right after, which is the state the reference
// reaches when its re-queue loop has settled.  time.Now() is a parameter.
#pragma once
#include <cstdint>
#include <memory>
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
    // getConnectionInfo: the fd's link names a socket inode; the first line of net/tcp that contains the inode's
    // digits is taken for the connection; all values are dropped and one open value stamped now_kernel_ns is added
    enum class Seed { Ok = 0, NoLink = 1, NoInode = 2, NoTcpFile = 3, NoLineFound = 4, ShortLine = 5 };
    Seed SeedFromProc(const std::string& proc_root, uint64_t now_kernel_ns);
    uint32_t Pid() const { return pid_; }
    uint64_t Fd() const { return fd_; }
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

namespace procfs {
// the digits of the first `socket:[<digits>]` in a /proc/<pid>/fd/<fd> link text (sock_num_line.go:358-363); false if none
bool InodeOfLink(const std::string& link, std::string* inode);
// columns 1 and 2 of a /proc/net/tcp line: little-endian hex IPv4 + ':' + hex port (sock_num_line.go:332-349, 384-397).
// Out-of-syntax hex pairs read as 0, ports beyond 65535 as 0, as the reference's ignored ParseInt errors leave them; a signed
// one-digit pair ("-8": the reference prints the text "-8") keeps its low byte.  /proc never writes either.
bool ParseTcpLine(const std::string& line, uint32_t* laddr, uint16_t* lport, uint32_t* raddr, uint16_t* rport);
}  // namespace procfs

// clusterInfo.SocketMaps + the two aggregator routines that touch them
class ConnTracker {
public:
    // processTcpConnect; returns true if a value was added to a line
    bool ProcessTcpConnect(const tcp_state::TcpConnectEvent& e);
    // one tick of clearSocketLines: every line whose last value is an open socket is reported through
    // ds->PersistAliveConnection (resolution of UIDs is left to the data store — GraphDS does it on the
    // GPU from the IPs; FromType/ToType/UIDs stay empty here), then DeleteUnused.  Returns lines reported.
    size_t Sweep(int64_t now_ms, bool send_alive, datastore::DataStore* ds);
    SocketLine* Line(uint32_t pid, uint64_t fd);                       // borrowed; dies with ClearProc(pid) — in-process callers that hold mu_'s owner still
    std::shared_ptr<SocketLine> Share(uint32_t pid, uint64_t fd) const { return Find(pid, fd); }   // owning: survives ClearProc(pid) (what the C API hands out)
    size_t Lines() const;
    // NewSocketLine(fetch = true): lines created from now on are seeded from `root` ("" = off, the default: a library
    // must not read /proc of whatever pids a replay happens to carry).  The seeded value's stamp is
    // convertUserTimeToKernelTime(now) = first_kernel − (first_user − now) (data.go:1745-1747); now_user_ns = 0 means
    // the wall clock.
    void SetProcRoot(const std::string& root, uint64_t first_kernel_ns, uint64_t first_user_ns, uint64_t now_user_ns = 0);
    // clearProc: every line of the process is gone (process exit)
    size_t ClearProc(uint32_t pid);
    uint64_t SeedsOk() const { std::lock_guard<std::mutex> g(mu_); return seeds_ok_; }       // (ProcessTcpConnect counts them under mu_)
    uint64_t SeedsFailed() const { std::lock_guard<std::mutex> g(mu_); return seeds_failed_; }
private:
    std::shared_ptr<SocketLine> Find(uint32_t pid, uint64_t fd) const;
    mutable std::mutex mu_;
    std::unordered_map<uint32_t, std::unordered_map<uint64_t, std::shared_ptr<SocketLine>>> maps_;   // pid -> fd -> line
    std::vector<std::shared_ptr<SocketLine>> all_;                                                    // in order of creation
    std::string proc_root_;
    uint64_t first_kernel_ = 0, first_user_ = 0, now_user_ = 0, seeds_ok_ = 0, seeds_failed_ = 0;
};

}  // namespace alaz
