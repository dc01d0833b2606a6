// CPU test of alaz_amd/csrc/join_host.hpp: random churn of pod / service upserts and deletes (persist.go:55-71,
// 114-130) on dense and sparse IP distributions.  After every batch
//   * lookup(ip) on the mirror == what the two reference maps say (pod only / service only / both / unknown),
//   * replaying the dirty-word log onto a shadow copy reproduces the mirror (the log the engine ships is complete).
// Prints "ok <stats>" and returns 0, or the first mismatch.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../alaz_amd/csrc/join_host.hpp"

using namespace sgjoin;

static int check(const Table& t, const std::vector<u32>& shadow, const std::vector<u32>& probe) {
    for (u32 w = 0; w < t.L.words; w++) if (shadow[w] != t.blob[w]) { std::printf("shadow mismatch at word %u\n", w); return 1; }
    for (u32 ip : probe) {
        u32 pod = ~0u, svc = ~0u;
        t.lookup_pod_svc(ip, pod, svc);
        auto p = t.pod_ip.find(ip); auto s = t.svc_ip.find(ip);
        const u32 wp = p == t.pod_ip.end() ? ~0u : p->second, ws = s == t.svc_ip.end() ? ~0u : s->second;
        if (pod != wp || svc != ws) { std::printf("ip %08x: got pod %d svc %d, want %d %d\n", ip, (int)pod, (int)svc, (int)wp, (int)ws); return 1; }
    }
    return 0;
}

static int run(int mode, bool blocks, u32 n_ips, u32 max_blocks, unsigned seed) {
    std::mt19937 rng(seed);
    Layout L = Table::make_layout(n_ips, n_ips + 8, max_blocks);
    std::vector<u32> mem(L.words), shadow(L.words);
    Table t; t.init(L, mem.data(), blocks);
    auto gen_ip = [&](bool svc) -> u32 {
        if (mode == 0) return (svc ? 0xAC100001u : 0x0A000001u) + rng() % (svc ? n_ips / 3 + 1 : n_ips);         // dense ranges (the synthetic clusters)
        if (mode == 1) return svc ? (0x0A600000u + (rng() & 0xFFFFFu)) : (0x0AF40000u + ((rng() % 64) << 8) + rng() % 110);  // random /12 service CIDR, /24 per node
        return rng();                                                                                            // no structure at all
    };
    std::vector<u32> live;
    int n_both = 0;
    size_t full = 0, incr = 0, words = 0;
    for (int round = 0; round < 60; round++) {
        const int ops = round == 0 ? (int)n_ips : 200;
        std::vector<u32> touched;
        for (int k = 0; k < ops && t.pod_ip.size() + t.svc_ip.size() < n_ips; k++) {
            const bool svc = rng() % 3 == 0;
            const unsigned r = rng() % 10;
            if (r < 7 || live.empty()) {
                const u32 ip = gen_ip(svc), id = rng() % (n_ips + 8);
                if (!t.upsert(svc, ip, id)) { std::printf("upsert failed\n"); return 1; }
                t.set_kind(id, svc ? 2 : 1);
                live.push_back(ip); touched.push_back(ip);
            } else if (r < 8) {                       // same IP into the other map too ("both"; at most ip2cap / 2 = 512 of them are kept)
                if (++n_both > 400) continue;
                const u32 ip = live[rng() % live.size()];
                if (!t.upsert(rng() & 1, ip, rng() % (n_ips + 8))) return 1;
                touched.push_back(ip);
            } else {
                const u32 ip = live[rng() % live.size()];
                t.erase(rng() & 1, ip); touched.push_back(ip);
            }
        }
        // what the engine does at the next launch
        if (t.need_full) { if (t.rebuilds == 0) t.rebuild(); shadow.assign(t.blob, t.blob + L.words); t.uploaded_full(); full++; }
        else { std::vector<std::pair<u32, u32>> d; t.take_dirty(d); for (auto& kv : d) shadow[kv.first] = kv.second; incr++; words += d.size(); }
        for (int k = 0; k < 300; k++) touched.push_back(gen_ip(k & 1));
        for (int k = 0; k < 300 && !live.empty(); k++) touched.push_back(live[rng() % live.size()]);
        if (check(t, shadow, touched)) { std::printf("mode %d blocks %d round %d\n", mode, (int)blocks, round); return 1; }
    }
    std::vector<u32> all;
    for (auto& kv : t.pod_ip) all.push_back(kv.first);
    for (auto& kv : t.svc_ip) all.push_back(kv.first);
    if (check(t, shadow, all)) return 1;
    std::printf("ok mode %d blocks %d: ips %zu blocks_used %u l1 %u ck_n %u ck2_n %u full_uploads %zu incremental %zu (avg %.1f words) rebuilds %llu\n",
                mode, (int)blocks, t.n_ips(), t.blocks_used, t.l1_entries, t.ck_n, t.ck2_n, full, incr, incr ? (double)words / incr : 0.0, (unsigned long long)t.rebuilds);
    return 0;
}

// Level 1 grown INCREMENTALLY to its load limit (a failed l1_put used to drop a live /24 and leave a dangling entry
// that the next allocation handed to another /24): a small or empty first build, then single upserts across 10..50 random
// /24s; after every upsert-batch every IP must resolve to its own id.
static int run_growth(unsigned seed) {
    std::mt19937 rng(seed);
    const u32 n_ips = 4000;
    Layout L = Table::make_layout(n_ips, n_ips + 8, 64);
    std::vector<u32> mem(L.words), shadow(L.words);
    Table t; t.init(L, mem.data(), true);
    std::vector<u32> all;
    const u32 first = rng() % 3 == 0 ? 0 : 1 + rng() % 40;
    for (u32 k = 0; k < first; k++) { const u32 ip = 0x0A000000u + ((rng() % 4) << 8) + rng() % 250; t.upsert(false, ip, k); all.push_back(ip); }
    t.rebuild(); shadow.assign(t.blob, t.blob + L.words); t.uploaded_full();
    const u32 nblk = 10 + rng() % 41;
    std::vector<u32> blocks;
    for (u32 k = 0; k < nblk; k++) blocks.push_back(rng() & 0xFFFFFFu);
    u32 id = first;
    for (int round = 0; round < 12; round++) {
        for (int k = 0; k < 40 && id < n_ips; k++) {
            const u32 ip = (blocks[rng() % nblk] << 8) | (1 + rng() % 250);
            if (!t.upsert(rng() % 4 == 0, ip, id++)) { std::printf("growth: upsert failed\n"); return 1; }
            all.push_back(ip);
        }
        if (t.need_full) { shadow.assign(t.blob, t.blob + L.words); t.uploaded_full(); }
        else { std::vector<std::pair<u32, u32>> d; t.take_dirty(d); for (auto& kv : d) shadow[kv.first] = kv.second; }
        if (check(t, shadow, all)) { std::printf("growth seed %u round %d (blocks_used %u l1 %u rebuilds %llu)\n", seed, round, t.blocks_used, t.l1_entries, (unsigned long long)t.rebuilds); return 1; }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::atoi(argv[2]) == 1) {                       // growth mode: argv[1] seeds
        const int n = std::atoi(argv[1]);
        for (int sd = 0; sd < n; sd++) if (run_growth(1000 + sd)) return 1;
        std::printf("ok growth %d seeds\n", n);
        return 0;
    }
    const unsigned seed = argc > 1 ? (unsigned)std::atoi(argv[1]) : 1;
    int rc = 0;
    rc |= run(0, true, 15000, 512, seed);
    rc |= run(1, true, 8000, 256, seed + 1);
    rc |= run(2, true, 3000, 64, seed + 2);
    rc |= run(0, false, 15000, 2, seed + 3);
    rc |= run(0, true, 150000, 2048, seed + 4);
    return rc;
}
