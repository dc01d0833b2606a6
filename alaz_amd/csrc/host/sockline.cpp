// sockline.cpp — see sockline.hpp.
#include "sockline.hpp"

#include <algorithm>
#include <climits>
#include <cstring>
#include <ctime>
#include <fstream>
#include <unistd.h>

#include "graph_ds.hpp"   // FormatIPv4

namespace alaz {

static bool SamePair(const SockInfo& a, const SockInfo& b) {
    return a.Saddr == b.Saddr && a.Sport == b.Sport && a.Daddr == b.Daddr && a.Dport == b.Dport;
}

void SocketLine::AddValue(uint64_t timestamp, const SockInfo* info) {
    std::lock_guard<std::mutex> g(mu_);
    // an open identical to the line's last open is not recorded again (sock_num_line.go:71-78)
    if (!values_.empty() && values_.back().Open && info && SamePair(values_.back().Info, *info)) return;
    TimestampedSocket v; v.Timestamp = timestamp; v.Open = info != nullptr; if (info) v.Info = *info;
    // first entry whose stamp is >= the new one: equal stamps end up after the newcomer (:311-322)
    auto it = std::lower_bound(values_.begin(), values_.end(), timestamp,
                               [](const TimestampedSocket& x, uint64_t ts) { return x.Timestamp < ts; });
    values_.insert(it, v);
}

SockErr SocketLine::GetValue(uint64_t timestamp, uint64_t now_ns, SockInfo* out) {
    std::lock_guard<std::mutex> g(mu_);
    const size_t n = values_.size();
    if (n == 0) return SockErr::Empty;
    const size_t index = (size_t)(std::lower_bound(values_.begin(), values_.end(), timestamp,
                                   [](const TimestampedSocket& x, uint64_t ts) { return x.Timestamp < ts; }) - values_.begin());
    if (index == n) {                                   // after the last entry
        values_[n - 1].LastMatch = now_ns;
        if (!values_[n - 1].Open) {                      // ... which is a close: accept the open before it for one minute
            if (n >= 2 && values_[n - 2].Open && timestamp - values_[n - 2].Timestamp < 60ull * 1000000000ull) { *out = values_[n - 2].Info; return SockErr::Ok; }
            return SockErr::ClosedLast;
        }
        *out = values_[n - 1].Info;
        return SockErr::Ok;
    }
    if (index == 0) {                                    // at or before the first entry
        if (values_[0].Open) { *out = values_[0].Info; return SockErr::Ok; }
        return SockErr::NoSmaller;
    }
    TimestampedSocket& prev1 = values_[index - 1];
    if (!prev1.Open) {                                   // matched on a close: neighbours to the same destination may stand in
        if (index >= 2 && values_[index - 2].Open && values_[index].Open &&
            values_[index - 2].Info.Daddr == values_[index].Info.Daddr && values_[index - 2].Info.Dport == values_[index].Info.Dport) {
            *out = (timestamp - values_[index - 2].Timestamp < values_[index].Timestamp - timestamp) ? values_[index - 2].Info : values_[index].Info;
            return SockErr::Ok;
        }
        return SockErr::Closed;
    }
    prev1.LastMatch = now_ns;
    *out = prev1.Info;
    return SockErr::Ok;
}

void SocketLine::DeleteUnused() {
    std::lock_guard<std::mutex> g(mu_);
    if (values_.size() <= 1) return;
    // two opens in a row: the first one's close never arrived, keep the second.  The pass stops one short
    // of the end, so a last element that is not the second of such a pair is not carried over (:168-180).
    std::vector<TimestampedSocket> kept;
    size_t i = 0;
    while (i + 1 < values_.size()) {
        if (values_[i].Open && values_[i + 1].Open) { kept.push_back(values_[i + 1]); i += 2; }
        else { kept.push_back(values_[i]); i += 1; }
    }
    values_.swap(kept);
    uint64_t newest = 0;
    for (const auto& v : values_) if (v.LastMatch != 0 && v.LastMatch > newest) newest = v.LastMatch;
    const uint64_t keep_for = 5ull * 60ull * 1000000000ull;
    for (long k = (long)values_.size() - 1; k >= 1; k--) {
        if (!values_[k].Open && values_[k - 1].Open && values_[k - 1].LastMatch + keep_for < newest) {
            values_.erase(values_.begin() + (k - 1), values_.begin() + (k + 1));
            k--;
        }
    }
}

bool SocketLine::LastOpen(SockInfo* out) const {
    std::lock_guard<std::mutex> g(mu_);
    if (values_.empty() || !values_.back().Open) return false;
    *out = values_.back().Info;
    return true;
}
size_t SocketLine::Size() const { std::lock_guard<std::mutex> g(mu_); return values_.size(); }
TimestampedSocket SocketLine::At(size_t i) const { std::lock_guard<std::mutex> g(mu_); return values_.at(i); }

namespace tcp_state {
TcpConnectEvent DecodeWire(const uint8_t* r) {
    TcpConnectEvent e;
    std::memcpy(&e.Fd, r, 8); std::memcpy(&e.Timestamp, r + 8, 8); std::memcpy(&e.Type, r + 16, 4); std::memcpy(&e.Pid, r + 20, 4);
    std::memcpy(&e.SPort, r + 24, 2); std::memcpy(&e.DPort, r + 26, 2);
    e.SAddr = ((uint32_t)r[28] << 24) | ((uint32_t)r[29] << 16) | ((uint32_t)r[30] << 8) | r[31];   // tcp.go:241: "%d.%d.%d.%d" of bytes 0..3
    e.DAddr = ((uint32_t)r[44] << 24) | ((uint32_t)r[45] << 16) | ((uint32_t)r[46] << 8) | r[47];
    return e;
}
}  // namespace tcp_state

namespace procfs {

bool InodeOfLink(const std::string& link, std::string* inode) {
    static const std::string lead = "socket:[";
    for (size_t at = link.find(lead); at != std::string::npos; at = link.find(lead, at + 1)) {
        size_t b = at + lead.size(), e = b;
        while (e < link.size() && link[e] >= '0' && link[e] <= '9') e++;
        if (e > b && e < link.size() && link[e] == ']') { *inode = link.substr(b, e - b); return true; }
    }
    return false;
}

// Go's strconv.ParseInt(text, 16, 64) with the error dropped: 0 for anything that is not [+-]hexdigits, the extreme
// value on overflow (callers clamp or truncate afterwards)
static int64_t HexInt(const std::string& t) {
    size_t i = 0; bool neg = false;
    if (!t.empty() && (t[0] == '+' || t[0] == '-')) { neg = t[0] == '-'; i = 1; }
    if (i == t.size()) return 0;
    uint64_t v = 0; bool big = false;
    for (; i < t.size(); i++) {
        const char c = t[i]; unsigned d;
        if (c >= '0' && c <= '9') d = (unsigned)(c - '0'); else if (c >= 'a' && c <= 'f') d = (unsigned)(c - 'a') + 10; else if (c >= 'A' && c <= 'F') d = (unsigned)(c - 'A') + 10; else return 0;
        big = big || (v >> 60) != 0;
        v = v * 16 + d;
    }
    const uint64_t lim = neg ? (1ull << 63) : (1ull << 63) - 1;
    if (big || v > lim) return neg ? INT64_MIN : INT64_MAX;
    return neg ? (int64_t)(0 - v) : (int64_t)v;
}

static bool IsGoSpace(char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

bool ParseTcpLine(const std::string& line, uint32_t* laddr, uint16_t* lport, uint32_t* raddr, uint16_t* rport) {
    std::string col[3]; int n = 0; size_t i = 0;
    while (n < 3) {
        while (i < line.size() && IsGoSpace(line[i])) i++;
        if (i == line.size()) break;
        const size_t b = i; while (i < line.size() && !IsGoSpace(line[i])) i++;
        col[n++] = line.substr(b, i - b);
    }
    if (n < 3 || col[1].size() < 9 || col[2].size() < 9) return false;
    auto addr = [](const std::string& c) {                     // byte k of the text is the k-th LOWEST byte of the address
        uint32_t a = 0;
        for (int k = 0; k < 4; k++) a |= ((uint32_t)HexInt(c.substr(2 * k, 2)) & 0xFFu) << (8 * k);
        return a;
    };
    auto port = [](const std::string& c) { const int64_t p = HexInt(c.substr(9)); return (uint16_t)((p < 0 || p > 65535) ? 0 : p); };
    *laddr = addr(col[1]); *lport = port(col[1]); *raddr = addr(col[2]); *rport = port(col[2]);
    return true;
}

}  // namespace procfs

SocketLine::Seed SocketLine::SeedFromProc(const std::string& root, uint64_t now_kernel_ns) {
    char target[256];
    const std::string fdpath = root + "/" + std::to_string(pid_) + "/fd/" + std::to_string(fd_);
    const ssize_t got = ::readlink(fdpath.c_str(), target, sizeof target - 1);
    if (got < 0) return Seed::NoLink;
    std::string inode;
    if (!procfs::InodeOfLink(std::string(target, (size_t)got), &inode)) return Seed::NoInode;
    std::ifstream tcp(root + "/" + std::to_string(pid_) + "/net/tcp");
    if (!tcp.is_open()) return Seed::NoTcpFile;
    std::string row; bool hit = false;
    while (std::getline(tcp, row)) {
        if (row.size() >= 65536) break;                          // the reference's line scanner stops at a 64 KiB token
        if (!row.empty() && row.back() == '\r') row.pop_back();
        if (row.find(inode) != std::string::npos) { hit = true; break; }     // substring of the WHOLE line, whichever column
    }
    if (!hit) return Seed::NoLineFound;
    SockInfo si; si.Pid = pid_; si.Fd = fd_;
    if (!procfs::ParseTcpLine(row, &si.Saddr, &si.Sport, &si.Daddr, &si.Dport)) return Seed::ShortLine;
    { std::lock_guard<std::mutex> g(mu_); values_.clear(); }
    AddValue(now_kernel_ns, &si);
    return Seed::Ok;
}

std::shared_ptr<SocketLine> ConnTracker::Find(uint32_t pid, uint64_t fd) const {
    std::lock_guard<std::mutex> g(mu_);
    auto p = maps_.find(pid);
    if (p == maps_.end()) return nullptr;
    auto f = p->second.find(fd);
    return f == p->second.end() ? nullptr : f->second;
}
SocketLine* ConnTracker::Line(uint32_t pid, uint64_t fd) { return Find(pid, fd).get(); }
size_t ConnTracker::Lines() const { std::lock_guard<std::mutex> g(mu_); return all_.size(); }

void ConnTracker::SetProcRoot(const std::string& root, uint64_t first_kernel_ns, uint64_t first_user_ns, uint64_t now_user_ns) {
    std::lock_guard<std::mutex> g(mu_);
    proc_root_ = root; first_kernel_ = first_kernel_ns; first_user_ = first_user_ns; now_user_ = now_user_ns;
}

size_t ConnTracker::ClearProc(uint32_t pid) {
    std::lock_guard<std::mutex> g(mu_);
    auto p = maps_.find(pid);
    if (p == maps_.end()) return 0;
    const size_t n = p->second.size();
    maps_.erase(p);
    all_.erase(std::remove_if(all_.begin(), all_.end(), [pid](const std::shared_ptr<SocketLine>& l) { return l->Pid() == pid; }), all_.end());
    return n;
}

bool ConnTracker::ProcessTcpConnect(const tcp_state::TcpConnectEvent& e) {
    if (e.Type != tcp_state::kEstablished && e.Type != tcp_state::kClosed) return false;
    const uint32_t localhost = 0x7F000001u;
    if (e.SAddr == localhost || e.DAddr == localhost) return false;
    std::shared_ptr<SocketLine> line = Find(e.Pid, e.Fd);
    if (e.Type == tcp_state::kEstablished) {
        if (!line) {
            // seeded outside the tracker's lock (file system reads) and published afterwards, like the reference's
            // creation worker (socket.go:44-82); of two racing creators one line wins, the other is dropped unpublished
            auto fresh = std::make_shared<SocketLine>(e.Pid, e.Fd);
            std::string root; uint64_t fk, fu, now;
            { std::lock_guard<std::mutex> g(mu_); root = proc_root_; fk = first_kernel_; fu = first_user_; now = now_user_; }
            bool seeded = false, tried = false;
            if (!root.empty()) {
                if (now == 0) { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); now = (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec; }
                tried = true; seeded = fresh->SeedFromProc(root, fk - (fu - now)) == SocketLine::Seed::Ok;
            }
            std::lock_guard<std::mutex> g(mu_);
            std::shared_ptr<SocketLine>& slot = maps_[e.Pid][e.Fd];
            if (!slot) { slot = fresh; all_.push_back(fresh); if (tried) (seeded ? seeds_ok_ : seeds_failed_)++; }
            line = slot;
        }
        SockInfo si; si.Pid = e.Pid; si.Fd = e.Fd; si.Saddr = e.SAddr; si.Sport = e.SPort; si.Daddr = e.DAddr; si.Dport = e.DPort;
        line->AddValue(e.Timestamp, &si);
        return true;
    }
    if (!line) return false;
    line->AddValue(e.Timestamp, nullptr);
    return true;
}

size_t ConnTracker::Sweep(int64_t now_ms, bool send_alive, datastore::DataStore* ds) {
    std::vector<std::shared_ptr<SocketLine>> lines;
    { std::lock_guard<std::mutex> g(mu_); lines = all_; }
    size_t sent = 0;
    for (const auto& l : lines) {
        SockInfo si;
        if (send_alive && ds && l->LastOpen(&si)) {
            datastore::AliveConnection ac;
            ac.CheckTime = now_ms;
            ac.FromIP = FormatIPv4(si.Saddr); ac.FromPort = si.Sport;
            ac.ToIP = FormatIPv4(si.Daddr); ac.ToPort = si.Dport;
            ds->PersistAliveConnection(&ac);
            sent++;
        }
        l->DeleteUnused();
    }
    return sent;
}

}  // namespace alaz
