// sockline.cpp — see sockline.hpp.
#include "sockline.hpp"

#include <algorithm>
#include <cstring>

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

ConnTracker::~ConnTracker() { for (auto* l : all_) delete l; }

SocketLine* ConnTracker::Line(uint32_t pid, uint64_t fd) {
    std::lock_guard<std::mutex> g(mu_);
    auto p = maps_.find(pid);
    if (p == maps_.end()) return nullptr;
    auto f = p->second.find(fd);
    return f == p->second.end() ? nullptr : f->second;
}
size_t ConnTracker::Lines() const { std::lock_guard<std::mutex> g(mu_); return all_.size(); }

bool ConnTracker::ProcessTcpConnect(const tcp_state::TcpConnectEvent& e) {
    if (e.Type != tcp_state::kEstablished && e.Type != tcp_state::kClosed) return false;
    const uint32_t localhost = 0x7F000001u;
    if (e.SAddr == localhost || e.DAddr == localhost) return false;
    SocketLine* line = Line(e.Pid, e.Fd);
    if (e.Type == tcp_state::kEstablished) {
        if (!line) {
            std::lock_guard<std::mutex> g(mu_);
            SocketLine*& slot = maps_[e.Pid][e.Fd];
            if (!slot) { slot = new SocketLine(e.Pid, e.Fd); all_.push_back(slot); }
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
    std::vector<SocketLine*> lines;
    { std::lock_guard<std::mutex> g(mu_); lines = all_; }
    size_t sent = 0;
    for (SocketLine* l : lines) {
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
