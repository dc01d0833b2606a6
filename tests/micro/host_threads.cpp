// ThreadSanitizer driver for the C++ host side (tools/tsan_host.sh): eight feeder threads push perf records through
// sgh_graphds_ingest_wire while one thread churns the pod table, one feeds TCP events + sweeps and one flushes windows,
// all against the recording engine of host_capi.cpp.  Exit code 0 and no TSan report = pass.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/servicegraph.h"

extern "C" {
void* sgh_graphds_create(const char* engine_lib, const sg_config* cfg, size_t batch);
void sgh_graphds_destroy(void* g);
int sgh_graphds_persist_pod(void* g, const char* et, const char* uid, const char* ip);
int sgh_graphds_persist_service(void* g, const char* et, const char* uid, const char* ip);
int sgh_graphds_ingest_wire(void* g, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs);
long sgh_graphds_flush(void* g, int64_t window_end_ms, void* out, size_t cap);
size_t sgh_graphds_tcp_wire(void* g, const uint8_t* recs, size_t n);
size_t sgh_graphds_sweep(void* g, int64_t now_ms, int send_alive);
void sgh_graphds_proc_exec(void* g, uint32_t pid);
void sgh_graphds_proc_exit(void* g, uint32_t pid);
void sgh_graphds_set_proc_root(void* g, const char* root, uint64_t first_kernel_ns, uint64_t first_user_ns, uint64_t now_user_ns);
void sgh_graphds_sweep_http2(void* g);
size_t sgh_mock_events(void* g, sg_event* out, size_t cap);
}

static void l7(uint8_t* r, uint32_t pid, uint64_t fd, uint8_t proto, uint8_t method, const char* payload, uint32_t n, uint32_t saddr, uint32_t daddr, uint64_t wt) {
    std::memset(r, 0, 1096);
    std::memcpy(r, &fd, 8); std::memcpy(r + 8, &wt, 8); std::memcpy(r + 16, &pid, 4);
    uint32_t status = 200; std::memcpy(r + 20, &status, 4); uint64_t dur = 5000; std::memcpy(r + 24, &dur, 8);
    r[32] = proto; r[33] = method;
    std::memcpy(r + 36, payload, n); std::memcpy(r + 1060, &n, 4); r[1064] = 1;
    std::memcpy(r + 1076, &saddr, 4); std::memcpy(r + 1084, &daddr, 4);
}

int main() {
    sg_config cfg; std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg; cfg.abi_version = SG_ABI_VERSION; cfg.max_known_nodes = 256; cfg.max_edges = 4096;
    void* g = sgh_graphds_create(nullptr, &cfg, 64);
    if (!g) { std::fprintf(stderr, "create failed\n"); return 2; }
    for (int i = 0; i < 32; i++) { char uid[16], ip[16]; std::snprintf(uid, sizeof uid, "p%d", i); std::snprintf(ip, sizeof ip, "10.0.0.%d", i + 1); sgh_graphds_persist_pod(g, "ADD", uid, ip); }
    sgh_graphds_persist_service(g, "ADD", "s0", "10.96.0.1");
    sgh_graphds_set_proc_root(g, "/proc", 1000, 2000, 0);          // new socket lines try the proc file system first (whatever pid 77 is here)
    constexpr int kThreads = 8, kPer = 4000;
    std::atomic<bool> stop{false};
    std::vector<std::thread> ts;
    for (int t = 0; t < kThreads; t++) ts.emplace_back([&, t] {
        std::vector<uint8_t> buf(50 * 1096);
        for (int i = 0; i < kPer; i += 50) {
            for (int k = 0; k < 50; k++) {
                const int e = i + k; const uint32_t s = 0x0A000001u + (uint32_t)((t * 7 + e) % 32);
                if (e % 5 == 0) l7(&buf[k * 1096], 77, (uint64_t)t, 3, 2, "Q\0\0\0\x0dselect 1", 13, s, 0x0A600001u, 1000 + e);            // postgres
                else if (e % 7 == 0) l7(&buf[k * 1096], 77, (uint64_t)t, 1, 1, "GET / HTTP/1.1\r\nHost: ext.example\r\n\r\n", 37, s, 0x08080808u, 1000 + e);
                else l7(&buf[k * 1096], 77, (uint64_t)t, 1, 1, "GET /user HTTP1.1", 17, s, 0x0A600001u, 1000 + e);
            }
            sgh_graphds_ingest_wire(g, buf.data(), 50, nullptr);
        }
    });
    std::thread churn([&] { int k = 0; while (!stop) { char ip[20]; std::snprintf(ip, sizeof ip, "10.9.0.%d", k % 9 + 1); sgh_graphds_persist_pod(g, k & 1 ? "UPDATE" : "DELETE", "churn", ip); sgh_graphds_proc_exec(g, 1000 + k % 5); sgh_graphds_proc_exit(g, 1000 + (k + 2) % 5); if (k % 16 == 0) sgh_graphds_proc_exit(g, 77); k++; } });   // (77: the tcp thread's process — its lines go while it adds to them and the sweep walks them)
    std::thread tcp([&] {
        uint8_t r[64]; uint64_t ts_ = 1; 
        while (!stop) {
            std::memset(r, 0, 64); uint64_t fd = ts_ % 11; std::memcpy(r, &fd, 8); std::memcpy(r + 8, &ts_, 8);
            uint32_t type = ts_ % 3 ? 1 : 5, pid = 77; std::memcpy(r + 16, &type, 4); std::memcpy(r + 20, &pid, 4);
            r[28] = 10; r[31] = 1; r[44] = 10; r[45] = 96; r[47] = 1;
            sgh_graphds_tcp_wire(g, r, 1);
            if (ts_ % 64 == 0) { sgh_graphds_sweep(g, (int64_t)ts_, 1); sgh_graphds_sweep_http2(g); }
            ts_++;
        }
    });
    std::thread flusher([&] { while (!stop) sgh_graphds_flush(g, 1, nullptr, 0); });
    for (auto& t : ts) t.join();
    stop = true; churn.join(); tcp.join(); flusher.join();
    sgh_graphds_flush(g, 2, nullptr, 0);
    const size_t n = sgh_mock_events(g, nullptr, 0);
    std::vector<sg_event> ev(n); sgh_mock_events(g, ev.data(), n);
    size_t requests = 0; for (const auto& e : ev) requests += (e.flags & SG_EV_ALIVE) ? 0 : 1;
    sgh_graphds_destroy(g);
    std::printf("events recorded: %zu requests (want %d) + %zu alive\n", requests, kThreads * kPer, n - requests);
    return requests == (size_t)kThreads * kPer ? 0 : 1;
}
