// join_host.hpp — host side of the join tables (the build side of setFromToV2's join).
//
// The reference keeps two Go maps, PodIPToPodUid and ServiceIPToServiceUid (aggregator/cluster.go:13-17), written by
// processPod / processSvc with one map store or delete per k8s event (aggregator/persist.go:55-71, 114-130).  Here the
// authoritative copy is the same two maps (ip -> node id); what the kernels read is a flat array of 32-bit words
// ("blob") derived from them:
//
//   jl1   [l1_cap] u64   block table level 1: 2-choice hash of b = ip >> 8 -> {tag = b, blk}; tag all-ones = empty
//   jl2   [max_blocks][256] u32   level 2: kind << 30 | id per address of an allocated /24 block, 0 = unknown;
//                                 block 0 is never allocated and stays zero (misses are steered to it)
//   ck    [ipcap] u64    (2,2)-cuckoo table of the IPs whose /24 has no block ("residual"): ip | (kind << 30 | id) << 32
//   ck2   [ip2cap] u64   same, pod ids of IPs that are in BOTH maps (kind 3; the service id is in jl2 / ck)
//   kind  [max_known] u8 node kind per id (SG_NODE_POD / SG_NODE_SERVICE)
//
// Every mutation edits the host mirror AND logs the changed words; the engine ships the log to the device as a list of
// (word offset, value) pairs applied by a tiny kernel in stream order — an upsert costs O(1) words, not a table rebuild
// (the reference's map write is O(1) too).  A full rebuild + upload happens only when the log overflows, a table runs
// out of room, or the block assignment is stale (many IPs in /24s that have no block while blocks are free).
// No HIP in this file: tests/micro/join_host_test.cpp drives it on the CPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

#include "sg_hash.h"

namespace sgjoin {

typedef uint32_t u32;
typedef uint64_t u64;

inline u32 next_pow2_u32(u64 v) { u64 p = 1; while (p < v) p <<= 1; return (u32)p; }

struct Layout {
    u32 l1_cap = 0, max_blocks = 0, ipcap = 0, ip2cap = 0, max_known = 0;
    u32 off_l1 = 0, off_l2 = 0, off_ck = 0, off_ck2 = 0, off_kind = 0, words = 0;     // in 32-bit words, every section 16-byte aligned
};

class Table {
public:
    Layout L;
    u32* blob = nullptr;                         // host mirror, L.words words (owned by the caller: pinned memory in the engine)
    std::unordered_map<u32, u32> pod_ip, svc_ip; // authoritative: ip -> node id
    std::vector<std::pair<u32, u32>> dirty;      // (word offset, new value) since the last take_dirty()
    bool need_full = true;                       // the device copy must be replaced as a whole
    bool use_blocks = true;                      // false: every IP lives in the cuckoo table (K1 variant 1)
    u32 l1_entries = 16;                         // used prefix of jl1 (power of two)
    u32 blocks_used = 1;                         // block 0 = null block
    u32 ck_n = 0, ck2_n = 0;                     // IPs in ck / ck2
    size_t max_dirty = 1u << 15;
    u64 rebuilds = 0;

    static Layout make_layout(u32 max_ips, u32 max_known, u32 max_blocks) {
        Layout L;
        L.max_blocks = std::max<u32>(max_blocks, 2);
        L.l1_cap = next_pow2_u32((u64)4 * L.max_blocks);
        L.ipcap = next_pow2_u32(std::max<u64>((u64)max_ips * 5 / 4 + 1, 64));        // cuckoo 2x2: load factor <= 0.8
        L.ip2cap = 1024;
        L.max_known = max_known;
        auto al = [](u32 w) { return (w + 3u) & ~3u; };
        L.off_l1 = 0;
        L.off_l2 = al(L.off_l1 + 2 * L.l1_cap);
        L.off_ck = al(L.off_l2 + 256 * L.max_blocks);
        L.off_ck2 = al(L.off_ck + 2 * L.ipcap);
        L.off_kind = al(L.off_ck2 + 2 * L.ip2cap);
        L.words = al(L.off_kind + (max_known + 3) / 4);
        return L;
    }
    void init(const Layout& l, u32* mem, bool blocks) {
        L = l; blob = mem; use_blocks = blocks;
        kinds_.assign(L.max_known, 0);
        clear_blob();
        need_full = true;
    }

    // ---- mutations (persist.go:55-71, 114-130) -------------------------------------------------------------------
    // returns false when a table is out of room even after a rebuild (the engine reports SG_ENOSPC)
    bool upsert(bool svc, u32 ip, u32 id) {
        auto& m = svc ? svc_ip : pod_ip;
        auto it = m.find(ip);
        if (it != m.end() && it->second == id) return true;
        m[ip] = id;
        return place(ip);
    }
    bool erase(bool svc, u32 ip) {
        auto& m = svc ? svc_ip : pod_ip;
        if (!m.erase(ip)) return true;
        return place(ip);
    }
    void set_kind(u32 id, uint8_t k) {
        if (id >= L.max_known || kinds_[id] == k) return;
        kinds_[id] = k;
        const u32 w = L.off_kind + id / 4;
        u32 v = blob[w]; v &= ~(0xFFu << (8 * (id & 3))); v |= (u32)k << (8 * (id & 3));
        wr(w, v);
    }
    uint8_t kind_of(u32 id) const { return id < L.max_known ? kinds_[id] : 0; }
    size_t n_ips() const {                       // distinct IPs over both maps
        size_t both = 0;
        const auto& a = pod_ip.size() < svc_ip.size() ? pod_ip : svc_ip; const auto& b = pod_ip.size() < svc_ip.size() ? svc_ip : pod_ip;
        for (auto& kv : a) both += b.count(kv.first);
        return pod_ip.size() + svc_ip.size() - both;
    }

    // what the kernels compute, on the mirror: kind << 30 | id (kind 3: id = service), 0 = unknown
    u32 lookup(u32 ip) const {
        if (use_blocks) {
            const u32 b = ip >> 8, mask = l1_entries - 1;
            const u64 e1 = rd64(L.off_l1 + 2 * jl1_h1(b, mask)), e2 = rd64(L.off_l1 + 2 * jl1_h2(b, mask));
            const u32 blk = (u32)e1 == b ? (u32)(e1 >> 32) : ((u32)e2 == b ? (u32)(e2 >> 32) : 0u);
            const u32 v = blob[L.off_l2 + ((blk << 8) | (ip & 255u))];
            if (v) return v;
        }
        if (ck_n) { const u64 e = ck_find(L.off_ck, L.ipcap, ip); if (e != ~0ull) return (u32)(e >> 32); }
        return 0;
    }
    bool lookup_pod_svc(u32 ip, u32& pod, u32& svc) const {          // false: unknown IP
        const u32 v = lookup(ip), kind = v >> 30, id = v & 0x3FFFFFFFu;
        if (!v) return false;
        if (kind == 1) pod = id; else if (kind == 2) svc = id;
        else { svc = id; const u64 e2 = ck_find(L.off_ck2, L.ip2cap, ip); if (e2 != ~0ull) pod = (u32)(e2 >> 32) & 0x3FFFFFFFu; }
        return true;
    }

    // the log since the last call; need_full tells the caller to upload the whole blob instead
    void take_dirty(std::vector<std::pair<u32, u32>>& out) { out.swap(dirty); dirty.clear(); }
    void uploaded_full() { need_full = false; dirty.clear(); }

    // ---- full rebuild from the maps: blocks go to the most populated /24s -----------------------------------------
    bool rebuild() {
        rebuilds++;
        clear_blob();
        for (u32 id = 0; id < L.max_known; id++) if (kinds_[id]) blob[L.off_kind + id / 4] |= (u32)kinds_[id] << (8 * (id & 3));
        need_full = true; dirty.clear();
        if (use_blocks) {
            std::unordered_map<u32, u32> pop;
            for (auto& kv : pod_ip) pop[kv.first >> 8]++;
            for (auto& kv : svc_ip) if (!pod_ip.count(kv.first)) pop[kv.first >> 8]++;
            std::vector<std::pair<u32, u32>> order;                 // (population, block number)
            order.reserve(pop.size());
            for (auto& kv : pop) order.push_back({kv.second, kv.first});   // (singleton /24s too: the incremental path gives them blocks
                                                                           // as well, and a rebuild that cannot use the free blocks would be
                                                                           // triggered again and again by the stale-assignment rule)
            std::sort(order.begin(), order.end(), [](const std::pair<u32, u32>& a, const std::pair<u32, u32>& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
            const u32 take = (u32)std::min<size_t>(order.size(), L.max_blocks - 1);
            l1_entries = std::min<u32>(L.l1_cap, next_pow2_u32(std::max<u64>((u64)4 * (take + take / 4 + 8), 16)));   // load <= 0.25 with room to grow
            for (;;) {
                // below the cap one failed placement restarts with twice the slots; AT the cap a /24 that finds no slot is skipped and
                // the less populated ones behind it are still tried (stopping at the first failure left every remaining /24
                // in the cuckoo table, i.e. on K1's out-of-line general path, although most of them would fit)
                const bool at_cap = l1_entries >= L.l1_cap;
                bool placed = true;
                for (u32 i = 0; i < take && (placed || at_cap); i++) placed = alloc_block(order[i].second) && placed;
                if (placed || at_cap) break;                        // (at the cap the /24s that found no slot stay in the cuckoo table)
                // an insert walked 512 kicks at load <= 1/3: start over with twice the slots
                const u32 grown = l1_entries * 2;
                std::memset(blob + L.off_l1, 0xFF, (size_t)L.l1_cap * 8);
                for (u32 i = 0; i < L.l1_cap; i++) blob[L.off_l1 + 2 * i + 1] = 0;
                blk_of_.clear(); blocks_used = 1; l1_entries = grown;
            }
        }
        bool ok = true;
        for (auto& kv : pod_ip) ok &= place_fresh(kv.first);
        for (auto& kv : svc_ip) if (!pod_ip.count(kv.first)) ok &= place_fresh(kv.first);
        need_full = true; dirty.clear();
        return ok;
    }

    u32 blocks_bytes() const { return blocks_used * 1024u; }

private:
    std::vector<uint8_t> kinds_;
    std::unordered_map<u32, u32> blk_of_;        // block number -> level-2 block index (>= 1)
    std::unordered_map<u32, u32> ck_pop_;        // block number -> IPs of it in the cuckoo table
    u32 denied_since_rebuild_ = 0;

    void clear_blob() {
        std::memset(blob, 0, (size_t)L.words * 4);
        std::memset(blob + L.off_l1, 0xFF, (size_t)L.l1_cap * 8);
        for (u32 i = 0; i < L.l1_cap; i++) blob[L.off_l1 + 2 * i + 1] = 0;      // tag all-ones, blk 0
        std::memset(blob + L.off_ck, 0xFF, (size_t)L.ipcap * 8);
        std::memset(blob + L.off_ck2, 0xFF, (size_t)L.ip2cap * 8);
        blk_of_.clear(); ck_pop_.clear(); blocks_used = 1; ck_n = ck2_n = 0; denied_since_rebuild_ = 0;
        if (!use_blocks) l1_entries = 16;
    }
    void wr(u32 off, u32 v) {
        if (blob[off] == v) return;
        blob[off] = v;
        if (need_full) return;
        dirty.push_back({off, v});
        if (dirty.size() > max_dirty) { need_full = true; dirty.clear(); }
    }
    void wr64(u32 off, u64 v) { wr(off, (u32)v); wr(off + 1, (u32)(v >> 32)); }
    u64 rd64(u32 off) const { return (u64)blob[off] | ((u64)blob[off + 1] << 32); }

    u32 value_of(u32 ip, bool& both, u32& podid) const {
        auto p = pod_ip.find(ip); auto s = svc_ip.find(ip);
        both = false; podid = 0;
        if (p != pod_ip.end() && s != svc_ip.end()) { both = true; podid = p->second; return (3u << 30) | s->second; }
        if (s != svc_ip.end()) return (2u << 30) | s->second;
        if (p != pod_ip.end()) return (1u << 30) | p->second;
        return 0;
    }

    // ---- cuckoo tables (2 hash functions x 2 entries per bucket), random-walk eviction ----------------------------
    u64 ck_find(u32 off, u32 cap, u32 ip) const {
        const u32 bmask = (cap - 1) >> 1;
        for (u32 bb : {ip_h1(ip, bmask), ip_h2(ip, bmask)}) for (u32 s = 0; s < 2; s++) {
            const u64 e = rd64(off + 2 * (2 * bb + s));
            if (e != ~0ull && (u32)e == ip) return e;
        }
        return ~0ull;
    }
    bool ck_put(u32 off, u32 cap, u32 ip, u32 val) {
        const u32 bmask = (cap - 1) >> 1;
        u64 cur = (u64)ip | ((u64)val << 32);
        u32 avoid = 0xFFFFFFFFu, rng = sg_fmix32(ip) | 1u;
        for (u32 bb : {ip_h1(ip, bmask), ip_h2(ip, bmask)}) for (u32 s = 0; s < 2; s++) {      // an update must hit the existing entry, not a free slot before it
            const u32 o = off + 2 * (2 * bb + s);
            if (rd64(o) != ~0ull && blob[o] == ip) { wr64(o, cur); return true; }
        }
        for (int kick = 0; kick < 8192; kick++) {
            const u32 b1 = ip_h1((u32)cur, bmask), b2 = ip_h2((u32)cur, bmask);
            for (u32 bb : {b1, b2}) for (u32 s = 0; s < 2; s++) {
                const u32 o = off + 2 * (2 * bb + s);
                if (rd64(o) == ~0ull) { wr64(o, cur); return true; }
            }
            rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
            const u32 b = (b1 == avoid) ? b2 : (b2 == avoid ? b1 : ((rng & 2u) ? b2 : b1));
            const u32 o = off + 2 * (2 * b + (rng & 1u));
            const u64 ev = rd64(o); wr64(o, cur); cur = ev;
            avoid = b;
        }
        return false;                                                  // (the evicted key `cur` is lost: the caller rebuilds)
    }
    bool ck_del(u32 off, u32 cap, u32 ip) {
        const u32 bmask = (cap - 1) >> 1;
        for (u32 bb : {ip_h1(ip, bmask), ip_h2(ip, bmask)}) for (u32 s = 0; s < 2; s++) {
            const u32 o = off + 2 * (2 * bb + s);
            const u64 e = rd64(o);
            if (e != ~0ull && (u32)e == ip) { wr64(o, ~0ull); return true; }
        }
        return false;
    }

    // ---- block table ----------------------------------------------------------------------------------------------
    // Insert {b -> blk} into level 1 (2-choice, one entry per slot, random-walk eviction).  TRANSACTIONAL: when the walk does
    // not end within the kick budget every slot it touched is put back, so a failure costs nothing but the block (the caller
    // keeps the /24 in the cuckoo table) — it used to drop the last evicted, live /24 and leave a dangling entry behind.
    bool l1_put(u32 b, u32 blk) {
        const u32 mask = l1_entries - 1;
        u64 cur = (u64)b | ((u64)blk << 32);
        u32 rng = sg_fmix32(b) | 1u, avoid = 0xFFFFFFFFu;
        std::vector<std::pair<u32, u64>> undo;                       // (slot, what it held) in write order
        for (int kick = 0; kick < 512; kick++) {
            const u32 s1 = jl1_h1((u32)cur, mask), s2 = jl1_h2((u32)cur, mask);
            for (u32 s : {s1, s2}) if (blob[L.off_l1 + 2 * s] == (u32)cur) { wr64(L.off_l1 + 2 * s, cur); return true; }
            for (u32 s : {s1, s2}) if (blob[L.off_l1 + 2 * s] == SG_JL1_EMPTY) { wr64(L.off_l1 + 2 * s, cur); return true; }
            rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5;
            const u32 s = (s1 == avoid) ? s2 : (s2 == avoid ? s1 : ((rng & 1u) ? s2 : s1));
            const u64 ev = rd64(L.off_l1 + 2 * s);
            undo.push_back({s, ev});
            wr64(L.off_l1 + 2 * s, cur); cur = ev;
            avoid = s;
        }
        for (size_t i = undo.size(); i-- > 0;) wr64(L.off_l1 + 2 * undo[i].first, undo[i].second);
        return false;
    }
    bool alloc_block(u32 b) {
        if (blocks_used >= L.max_blocks) return false;
        if ((u64)3 * (blk_of_.size() + 1) > (u64)l1_entries) return false;         // level 1 beyond load 1/3 (2-choice single-slot cuckoo
                                                                                   // gives out at 1/2): grows at the next rebuild
        if (!l1_put(b, blocks_used)) return false;
        blk_of_[b] = blocks_used++;
        return true;
    }

    // (re)place one IP according to the maps; incremental path
    bool place(u32 ip) {
        if (need_full && rebuilds == 0) return true;                   // nothing built yet: the first sync rebuilds from the maps
        bool both; u32 podid;
        const u32 v = value_of(ip, both, podid);
        const u32 b = ip >> 8;
        bool ok = true;
        auto bi = use_blocks ? blk_of_.find(b) : blk_of_.end();
        if (bi == blk_of_.end() && use_blocks && v && !ck_pop_.count(b) && alloc_block(b)) bi = blk_of_.find(b);
        if (bi != blk_of_.end()) wr(L.off_l2 + ((bi->second << 8) | (ip & 255u)), v);
        else {
            const bool had = ck_find(L.off_ck, L.ipcap, ip) != ~0ull;
            if (v) {
                ok = ck_put(L.off_ck, L.ipcap, ip, v);
                if (ok && !had) {
                    ck_n++; ck_pop_[b]++;
                    // stale block assignment: many IPs went to the cuckoo table while blocks are free -> re-select
                    if (use_blocks && ++denied_since_rebuild_ > std::max<size_t>(256, (pod_ip.size() + svc_ip.size()) / 8) && blocks_used < L.max_blocks) ok = false;
                }
            } else if (had) {
                ck_del(L.off_ck, L.ipcap, ip); ck_n--;
                auto cp = ck_pop_.find(b); if (cp != ck_pop_.end() && --cp->second == 0) ck_pop_.erase(cp);
            }
        }
        // the pod id of an IP that is in both maps
        const bool had2 = ck2_n && ck_find(L.off_ck2, L.ip2cap, ip) != ~0ull;
        if (both) {
            if (!had2 && ck2_n >= L.ip2cap / 2) { /* second table full: the service mapping wins as destination, the pod side is lost */ }
            else { if (ck_put(L.off_ck2, L.ip2cap, ip, (1u << 30) | podid)) { if (!had2) ck2_n++; } else ok = false; }
        } else if (had2) { ck_del(L.off_ck2, L.ip2cap, ip); ck2_n--; }
        if (!ok) return rebuild();                                      // out of room / stale block assignment: start over from the maps
        return true;
    }
    // during rebuild(): tables are fresh, blocks already chosen
    bool place_fresh(u32 ip) {
        bool both; u32 podid;
        const u32 v = value_of(ip, both, podid);
        const u32 b = ip >> 8;
        bool ok = true;
        auto bi = blk_of_.find(b);
        if (bi != blk_of_.end()) blob[L.off_l2 + ((bi->second << 8) | (ip & 255u))] = v;
        else { ok = ck_put(L.off_ck, L.ipcap, ip, v); if (ok) { ck_n++; ck_pop_[b]++; } }
        if (both) {
            if (ck2_n < L.ip2cap / 2) { if (ck_put(L.off_ck2, L.ip2cap, ip, (1u << 30) | podid)) ck2_n++; else ok = false; }
        }
        return ok;
    }
};

}  // namespace sgjoin
