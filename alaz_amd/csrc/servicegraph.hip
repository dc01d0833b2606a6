// servicegraph.hip — host engine + C ABI (include/servicegraph.h) of the MI355X ServiceGraph engine.
//
// One sg_engine owns one device: the join tables, the open window's edge table, the closed
// window's CSR / feature / score buffers, a pinned staging ring for host-fed events and one HIP
// stream.  All kernels live in sg_kernels.h and the per-stage headers it includes.  There is no CPU compute path in this file: every
// entry point that produces results launches HIP kernels, and sg_create() fails without a device.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>

#include "join_host.hpp"
#include "sg_kernels.h"
#include "shard_seq.hpp"

namespace {

constexpr int kStageSlots = 32;                      // staging ring slots at most (sg_engine::n_stage of them are allocated)
constexpr int kUpdSlots = 4;             // pinned ring for join-table word updates
constexpr u32 kUpdCap = 1u << 15;         // (word offset, value) pairs per slot = sgjoin::Table::max_dirty
constexpr size_t kLdsBytes = 160 * 1024;  // per workgroup on gfx950

struct TimingRec { hipEvent_t a, b; int kernel; };

u32 next_pow2(u64 v) { u64 p = 1; while (p < v) p <<= 1; return (u32)p; }

// Tuning knobs (SG_NP, SG_HT, SG_K1A, SG_ABLATE, ...: tools/k1_sweep.py, the A/B tests of alternative kernel paths) are read from the
// environment by the DEVELOPMENT build only (-DSG_DEV_KNOBS -> lib/libservicegraph_dev.so); the shipped library reads none.
inline const char* sg_knob(const char* name) {
#ifdef SG_DEV_KNOBS
    return std::getenv(name);
#else
    (void)name; return nullptr;
#endif
}

}  // namespace

struct sg_engine {
    sg_config cfg{};
    std::mutex mu;
    std::string err;
    hipStream_t stream = nullptr;
    Dev d{};
    std::vector<void*> allocs;
    // SG_ARENA=1: everything below 32 MiB is carved out of 256 MiB chunks instead of one hipMalloc per array (a test of whether the
    // small arrays' page-table entries cost the window close — a chain of small latency-bound kernels — anything: measured on one box,
    // 507.7 / 508.3 us per C3 window with the chunks, 505.1 / 504.8 without; off).
    char* arena_base = nullptr; size_t arena_left = 0; bool arena_on = false;

    // join tables: the two reference maps + the word image the kernels read (join_host.hpp).  Mutations are logged as
    // changed words and shipped to the device copy in stream order (k_join_apply); a whole-image upload only when
    // the log overflows or a table was rebuilt.
    sgjoin::Table jt;
    std::vector<u32> jt_mirror;                      // host mirror (jt.blob)
    u32* h_blob = nullptr; u32* d_blob = nullptr;    // pinned staging of a whole-image upload, device copy
    hipEvent_t blob_ev = nullptr;                    // the last whole-image upload has finished reading h_blob
    uint2* h_upd[kUpdSlots] = {}; uint2* d_upd[kUpdSlots] = {}; hipEvent_t upd_ev[kUpdSlots] = {}; int upd_next = 0;
    hipEvent_t tab_ev = nullptr, k1_ev = nullptr;    // cross-stream ordering: table modification <-> K1 launches
    u64 tab_seq = 0;                                 // table modifications so far
    std::vector<std::pair<hipStream_t, u64>> seen_seq;   // per stream: the modification its K1 launches have been ordered after
    std::vector<hipStream_t> k1_streams;             // streams with K1 launches since the last table modification
    hipStream_t tab_stream = nullptr;
    u32 n_known = 0;
    // pass-A launch geometry, follows the table state
    bool l2_in_lds = false, l2_u16 = false, k1a_team = false; u32 k1a_ct = 2048, k1a_nsub = 2, k1a_teams = 2, k1a_nt = 1024;
    // staging ring for sg_ingest()
    sg_event* h_stage[kStageSlots] = {}; sg_event* d_stage[kStageSlots] = {}; hipEvent_t stage_ev[kStageSlots] = {};
    hipStream_t copy_stream = nullptr, copy_stream2 = nullptr; int n_copy = 1, copy_rr = 0; hipEvent_t copied_ev[kStageSlots] = {};   // H2D copies run on their own stream: batch i + 1 is copied while K1a folds batch i
    std::vector<std::pair<const char*, size_t>> registered;          // caller memory page-locked by sg_host_register
    int stage_next = 0, n_stage = 16;
    sg_edge_out* h_rows = nullptr; sg_edge_out* h_rows_old = nullptr; size_t h_rows_cap = 0;              // page-locked destination of sg_flush_window_view (grown on demand)
    bool stage_busy[kStageSlots] = {};                                   // a feeder thread is copying into the slot (outside the lock)
    int pending_copies = 0; std::condition_variable cv;                 // window closes wait for the copies that began before them
    // sg_flush_begin .. sg_flush_end: the window is closed and its pipeline enqueued under the lock (feeders that arrive meanwhile wait
    // on `closing`); the rows are fetched WITHOUT it, on rd_stream, while the feeders already fill the next window
    bool closing = false, flush_open = false, flush_async = false, fetching = false;
    hipStream_t rd_stream = nullptr; hipEvent_t score_ev = nullptr; u64* h_ctr_pin = nullptr;
    const sg_edge_out* fl_rows = nullptr; const u32* fl_ob = nullptr;    // device rows / outbound list of the window being flushed

    u64 first_kernel = 0, first_user = 0;
    float* d_W = nullptr; bool have_w = false;
    u32 n_labels_decl = 0;
    u32 ecap = 0, obcap = 0, ob_list_cap = 0;
    u32* d_ob_list = nullptr; u32* d_ob_n = nullptr;

    sg_stats st{};
    u64 h_ctr[C_COUNT] = {};
    std::vector<u32> last_obips;
    bool closed = false;       // window_close has run; rows readable after score
    bool use_mfma = true;
    int k1_grid = 0;
    size_t k1a_lds = 0, k1b_lds = 0, k3in_lds = 0;
    u32 k3_ranges = 1, k3_slices = 8;
    u32 k1b_threads = 512, k1b_u = 4, k1b_cus = 256;
    bool k1b_pack = false;                    // narrow pass B: count + duration sum of a record in one 64-bit LDS add (sn x nwg < 2^16)
    // warm windows (sg_device.h): the host only decides whether a window TRIES the warm path; whether it may is decided on the device
    bool warm_on = true;                      // sg_set_warm
    u32 cold_streak = 0;                      // consecutive windows whose warm attempt met an unknown key (C_COLD = 2), by the device's note
    u32 kw_epoch = 0;                         // kw_compact launch counter (tags its look-back words)
    u32 in_fused_slices = 0;                  // != 0 between a one-call pipeline's close and its features: the features sum k3_in_part's partials (k3_in_reduce was not launched)
    u64 closes = 0; u32 timing_stride = 1;    // windows closed so far; the dispatch stamps of K1 (groups 1 and 7) are taken on every timing_stride-th window
    u32 obip_streak = 0; u64 plain_left = 0;  // windows in a row that raw outbound IPs kept cold (same note); windows still to be closed without the kept-CSR detour
    std::vector<char> plain_slot;             // per window slot: its last close was such a plain one (not counted as warm or cold)
    std::vector<u64*> h_note; std::vector<u64> note_seen;   // per window slot: the device's note (page-locked, mapped) and the sequence number last read from it
    std::vector<u64*> scr_sum, scr_max; std::vector<double*> scr_mu;   // per window slot: the node statistics the kept-CSR rebuild writes (scratch; row_mu | row_sd in one array)
    // (Measured and rejected, profiles/r05_c_aux*: the rebuild kernels on a second stream beside the warm window's compaction, so that their
    // empty launches cost nothing — the fork / join events and the two streams' kernels slowing each other cost more: 506 vs 490 us per window.)
    u64 window_events_in = 0;

    unsigned timing = 0;       // bit k set: kernel group k is bracketed by HIP events
    hipEvent_t win_ta = nullptr;   // timing group 10: the open window's begin event (recorded in front of its first pass-A launch)
    u32 k6_epoch = 0;          // k6_halo_lists launch counter (same use)
    u32 rp_epoch = 0;          // k2_rowptr launch counter (tags the per-workgroup totals, so they need no reset)

    // windows in flight: every slot has its own window buffers and stream; the members above (d, stream,
    // d_ob_list, d_ob_n, closed, window_events_in) are the working copy of slot `cur`
    struct WinSlot { Dev d; hipStream_t stream; u32* ob_list; u32* ob_n; bool closed; u64 events_in; };
    std::vector<WinSlot> slots;
    int cur = 0;
    sg_edge_out* last_rows = nullptr;
    std::vector<TimingRec> trecs;
    std::vector<hipEvent_t> ev_pool;
    // sg_window_run_sharded: exchange buffers of the sharded window (allocated at the first call; world = cfg.world)
    struct Xchg { u32* ob_local = nullptr; u32* ob_all = nullptr; u32* req = nullptr; u32* serve = nullptr; float* rows_out = nullptr; float* rows_in = nullptr; u32 capp = 0, ob_stride = 0; } xc;
};

namespace {

#define HIP_TRY(e, call)                                                                       \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) {                                                                \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_r);                      \
            return _r == hipErrorOutOfMemory ? SG_ENOMEM : SG_ENODEV;                          \
        }                                                                                      \
    } while (0)

template <typename T>
int dev_alloc(sg_engine* e, T** p, size_t n, int fill = 0) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    constexpr size_t kArenaMax = (size_t)32 << 20, kArenaChunk = (size_t)256 << 20, kArenaAlign = 4096;
    if (e->arena_on && bytes < kArenaMax) {
        const size_t need = (bytes + kArenaAlign - 1) & ~(kArenaAlign - 1);
        if (need > e->arena_left) {                                  // (what is left of the previous chunk stays unused)
            void* c = nullptr;
            HIP_TRY(e, hipMalloc(&c, kArenaChunk));
            e->allocs.push_back(c);
            e->arena_base = static_cast<char*>(c); e->arena_left = kArenaChunk;
        }
        q = e->arena_base; e->arena_base += need; e->arena_left -= need;
    } else {
        HIP_TRY(e, hipMalloc(&q, bytes));
        e->allocs.push_back(q);
    }
    HIP_TRY(e, hipMemsetAsync(q, fill, bytes, e->stream));
    *p = (T*)q;
    return SG_OK;
}

hipStream_t pick(sg_engine* e, void* s) { return s ? (hipStream_t)s : e->stream; }

hipEvent_t get_event(sg_engine* e) {
    if (!e->ev_pool.empty()) { hipEvent_t v = e->ev_pool.back(); e->ev_pool.pop_back(); return v; }
    hipEvent_t v; hipEventCreate(&v); return v;
}

struct Timed {
    sg_engine* e; hipStream_t s; TimingRec r; bool on;
    Timed(sg_engine* e_, hipStream_t s_, int kernel) : e(e_), s(s_), on(e_ && ((e_->timing >> kernel) & 1u)) {
        if (on) { r.kernel = kernel; r.a = get_event(e); r.b = get_event(e); hipEventRecord(r.a, s); }
    }
    ~Timed() { if (on) { hipEventRecord(r.b, s); e->trecs.push_back(r); } }
};

// ---- join tables on the device: the host mirror's changes since the last launch, in stream order ------------------
// (processPod / processSvc analogue: aggregator/persist.go:55-71, 114-130 — one map write there, a few words here)

// pass-A geometry that follows the table state: does level 2 fit LDS beside the edge cache (and, narrow path, the tile)?
// Returns false when no legal geometry exists (the caller fails the call with a message instead of launching a kernel that
// asks for more than a CU's LDS).
bool k1a_geometry(sg_engine* e) {
    const Dev& d = e->d;
    const size_t l1b = (size_t)e->jt.l1_entries * 8, l2b = (size_t)e->jt.blocks_bytes();
    const size_t stage_max = (size_t)K1A_NJ * K1A_THREADS * 16;    // what the prologue can stage: six 16-byte words per lane
    e->l2_in_lds = false;
    if (d.narrow) {
        // cache | 6 counters per partition | statistics | tile | join tables.  Level 2 is staged as u16 entries
        // (half the bytes) when every node id fits 14 bits.
        e->l2_u16 = e->cfg.max_known_nodes <= 16384 && !sg_knob("SG_L2_U32");
        const size_t l2lds = e->l2_u16 ? l2b / 2 : l2b;
        if (const char* v = sg_knob("SG_NSUB")) { const int x = std::atoi(v); if (x == 1 || x == 2) e->k1a_nsub = (u32)x; }
        // round 4: two teams per workgroup (k1a_team_partition) when their two tiles and counter sets fit beside a cache of 256 slots or
        // more; SG_K1A=tile keeps the one-team kernel (k1a_tile_partition), SG_K1A=team takes the two-team kernel whenever it fits at all
        const char* kv = sg_knob("SG_K1A");
        const bool want_tile = kv && !std::strcmp(kv, "tile"), force_team = kv && !std::strcmp(kv, "team");
        const size_t fixed_tile = (size_t)d.np * 24 + 64 + (size_t)K1T_TS(e->k1a_nsub) * 8 + l1b;
        // the two-team kernel is instantiated for 256 / 512 / 1024 partitions, 1024 threads (two teams of eight waves, one group of four events
        // per thread and tile: round 6; round 4 ran two groups at 768 threads) and a join blob its prologue can stage; everything else keeps
        // the one-team kernel
        const int teams = 2, nt = 1024;
        const size_t lds_cap = kLdsBytes;
        // (what the team kernel's prologue can stage: six 16-byte words per lane of its threads.  Level 1 always goes through it, level 2
        // only when it is staged — an engine whose level 2 stays in global memory needs room for level 1 alone)
        const size_t stage_team = (size_t)K1A_NJ * nt * 16;
        // (endpoint spaces beyond 15 bits — a shard of BASELINE config 5: 18 — leave no room for the partition number in a parked record: the
        // kernel's 16-lanes-per-run copy-out takes over, the workgroup has the 1024 threads it is written for since round 6)
        const bool team_ok = (d.np == 256 || d.np == 512 || d.np == 1024) && (size_t)d.np * d.nwg * d.punits * 8 < ((size_t)1 << 31) && l1b <= stage_team;
        e->k1a_teams = (u32)teams; e->k1a_nt = (u32)nt;
        const size_t fixed_team = K1M_LDS_FIXED(d.np, teams, nt) + l1b;
        auto pick = [&](size_t fixed, u32 ct_min, size_t stage_cap, u32& ct, bool& in_lds) {
            ct = 0; in_lds = false;
            // (the cache flattens the hottest keys; beyond 1024 slots it costs more aggregates than it saves records)
            for (u32 c : {1024u, 512u, 256u, 128u}) if (c >= ct_min && (size_t)c * 40 + fixed + l2lds <= lds_cap && l1b + l2b <= stage_cap) { in_lds = true; ct = c; break; }
            if (!ct) for (u32 c : {1024u, 512u, 256u, 128u, 64u}) if (c >= ct_min && (size_t)c * 40 + fixed <= lds_cap) { ct = c; break; }
        };
        u32 ct = 0; bool in_lds = false;
        e->k1a_team = false;
        if (!want_tile && team_ok) { pick(fixed_team, force_team ? 64u : 256u, stage_team, ct, in_lds); e->k1a_team = ct != 0; }
        if (!ct) pick(fixed_tile, 64u, stage_max, ct, in_lds);
        const size_t fixed = e->k1a_team ? fixed_team : fixed_tile;
        e->l2_in_lds = in_lds;
        if (sg_knob("SG_L2_GLOBAL")) e->l2_in_lds = false;
        if (const char* v = sg_knob("SG_CT")) { const u32 x = (u32)std::strtoul(v, nullptr, 0); if (x >= 64 && x <= 2048 && (x & (x - 1)) == 0 && (size_t)x * 40 + fixed + (e->l2_in_lds ? l2lds : 0) <= lds_cap) ct = x; }
        if (!ct || l1b > stage_max) return false;
        e->k1a_ct = ct;
        e->k1a_lds = (size_t)ct * 40 + fixed + (e->l2_in_lds ? l2lds : 0);
        return true;
    }
    const size_t fixed = (size_t)d.np * 4 + 64 + l1b;
    const size_t slot = d.hist ? 72 : 40;                    // cache slot: key + 4 accumulators (+ 16 x u16 bins)
    u32 ct = 0;
    for (u32 c : {2048u, 1024u, 512u}) if ((size_t)c * slot + fixed + l2b <= kLdsBytes && l1b + l2b <= stage_max) { e->l2_in_lds = true; ct = c; break; }
    // level 2 stays in global memory: the largest cache that fits beside the counters and level 1
    if (!ct) for (u32 c : {2048u, 1024u, 512u, 256u, 128u, 64u}) if ((size_t)c * slot + fixed <= kLdsBytes) { ct = c; break; }
    if (const char* v = sg_knob("SG_CT")) { const u32 x = (u32)std::strtoul(v, nullptr, 0); if (x >= 64 && x <= 2048 && (x & (x - 1)) == 0 && (size_t)x * slot + fixed + (e->l2_in_lds ? l2b : 0) <= kLdsBytes) ct = x; }
    if (sg_knob("SG_L2_GLOBAL")) e->l2_in_lds = false;
    if (!ct || l1b > stage_max) return false;
    e->k1a_ct = ct;
    e->k1a_lds = (size_t)ct * slot + fixed + (e->l2_in_lds ? l2b : 0);
    return true;
}
// the sizes a K1 launch needs from the table state (pointers are fixed at create)
void join_view(const sg_engine* e, Dev& d) {
    d.jl1mask = e->jt.l1_entries - 1;
    d.jl2_words = e->jt.use_blocks ? e->jt.blocks_used * 256u : 0u;
    d.ck_n = e->jt.ck_n;
    d.jstage_bytes = (u32)((size_t)e->jt.l1_entries * 8 + (e->l2_in_lds ? e->jt.blocks_bytes() : 0));
    d.jl2_in_lds = e->l2_in_lds ? 1u : 0u;
    d.k1a_ct = e->k1a_ct;
}

int sync_tables(sg_engine* e, hipStream_t s) {
    sgjoin::Table& jt = e->jt;
    if (!jt.need_full && jt.dirty.empty()) return SG_OK;
    // the modification runs on stream s: after every K1 launch that may still read the tables on another stream
    for (hipStream_t ks : e->k1_streams) if (ks != s) { HIP_TRY(e, hipEventRecord(e->k1_ev, ks)); HIP_TRY(e, hipStreamWaitEvent(s, e->k1_ev, 0)); }
    e->k1_streams.clear();
    if (jt.rebuilds == 0 && !jt.rebuild()) { e->err = "join table build failed: raise max_ips"; return SG_ENOSPC; }
    if (jt.need_full) {
        HIP_TRY(e, hipEventSynchronize(e->blob_ev));                 // the previous whole-image upload has finished reading h_blob
        std::memcpy(e->h_blob, jt.blob, (size_t)jt.L.words * 4);
        HIP_TRY(e, hipMemcpyAsync(e->d_blob, e->h_blob, (size_t)jt.L.words * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(e, hipEventRecord(e->blob_ev, s));
        jt.uploaded_full();
        e->st.join_full_uploads++;
    } else {
        std::vector<std::pair<u32, u32>> log;
        jt.take_dirty(log);
        // one pair per word (the last value wins): the apply kernel writes them in parallel
        std::unordered_map<u32, u32> last;
        last.reserve(log.size() * 2);
        for (auto& kv : log) last[kv.first] = kv.second;
        const int slot = e->upd_next; e->upd_next = (slot + 1) % kUpdSlots;
        HIP_TRY(e, hipEventSynchronize(e->upd_ev[slot]));
        u32 n = 0;
        for (auto& kv : last) e->h_upd[slot][n++] = make_uint2(kv.first, kv.second);
        HIP_TRY(e, hipMemcpyAsync(e->d_upd[slot], e->h_upd[slot], (size_t)n * sizeof(uint2), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_join_apply, dim3((n + 255) / 256), dim3(256), 0, s, e->d_blob, (const uint2*)e->d_upd[slot], n);
        HIP_TRY(e, hipEventRecord(e->upd_ev[slot], s));
        e->st.join_word_updates += n;
    }
    HIP_TRY(e, hipEventRecord(e->tab_ev, s));
    e->tab_seq++; e->tab_stream = s;
    if (e->d.variant == 0 && !k1a_geometry(e)) { e->err = "K1 pass A: the join tables' level 1 and the piece counters do not fit a CU's LDS (fewer partitions / IP blocks needed)"; return SG_ENOSPC; }
    return SG_OK;
}
// K1 on stream s reads the tables: after the last modification if that ran on another stream
int order_after_tables(sg_engine* e, hipStream_t s) {
    if (e->tab_seq) {
        u64* seen = nullptr;
        for (auto& kv : e->seen_seq) if (kv.first == s) seen = &kv.second;
        if (!seen) { e->seen_seq.push_back({s, 0}); seen = &e->seen_seq.back().second; }
        if (*seen != e->tab_seq) { if (e->tab_stream != s) HIP_TRY(e, hipStreamWaitEvent(s, e->tab_ev, 0)); *seen = e->tab_seq; }
    }
    if (std::find(e->k1_streams.begin(), e->k1_streams.end(), s) == e->k1_streams.end()) e->k1_streams.push_back(s);
    return SG_OK;
}

// make slot (cur + 1) % NW the working window; the closed window keeps running on its own stream
void rotate_window(sg_engine* e) {
    if (e->slots.size() < 2) return;
    sg_engine::WinSlot& a = e->slots[e->cur];
    a.d = e->d; a.stream = e->stream; a.ob_list = e->d_ob_list; a.ob_n = e->d_ob_n; a.closed = e->closed; a.events_in = e->window_events_in;
    e->cur = (e->cur + 1) % (int)e->slots.size();
    const sg_engine::WinSlot& b = e->slots[e->cur];
    e->d = b.d; e->stream = b.stream; e->d_ob_list = b.ob_list; e->d_ob_n = b.ob_n; e->closed = b.closed; e->window_events_in = b.events_in;
}

int launch_k1(sg_engine* e, const sg_event* d_ev, size_t n, hipStream_t s) {
    if (n == 0) return SG_OK;
    // the pass-A cache keeps 16-bit histogram bins: a workgroup must see fewer than 65536 events per launch
    const size_t kMaxHist = (size_t)65535 * 256;
    if (e->d.hist && e->d.variant == 0 && n > kMaxHist) {
        for (size_t o = 0; o < n; o += kMaxHist) { const int rc = launch_k1(e, d_ev + o, std::min(kMaxHist, n - o), s); if (rc) return rc; }
        return SG_OK;
    }
    const size_t kMaxTeam = (size_t)1 << 25;                             // k1a_team_partition addresses a launch's events with byte offsets below 2 GiB
    if (e->d.variant == 0 && e->d.narrow && e->k1a_team && n > kMaxTeam) {
        for (size_t o = 0; o < n; o += kMaxTeam) { const int rc = launch_k1(e, d_ev + o, std::min(kMaxTeam, n - o), s); if (rc) return rc; }
        return SG_OK;
    }
    int rc = sync_tables(e, s);
    if (rc) return rc;
    if ((rc = order_after_tables(e, s))) return rc;
    if (((e->timing >> 10) & 1u) && e->window_events_in == 0 && !e->win_ta) { e->win_ta = get_event(e); hipEventRecord(e->win_ta, s); }   // group 10: the whole window
    Dev da = e->d;
    join_view(e, da);
    u32 rot0 = e->d.k1a_rot, tb0 = e->d.k1a_ticket_base;
    if (e->d.variant == 0) {
        // single-kernel groups are timed by the dispatch's own begin/end stamps (hipExtLaunchKernel start/stop
        // events): the kernel's duration as rocprofv3 reports it, without the event-record round trip
        const bool tk = ((e->timing >> 1) & 1u) && e->closes % e->timing_stride == 0;   // (the window this batch belongs to is the closes-th)
        hipEvent_t ta = tk ? get_event(e) : nullptr, tb = tk ? get_event(e) : nullptr;
        da.batch_state = e->window_events_in == 0 ? 1u : 0u;     // first batch of this window?
        const bool sh = e->d.world > 1;
#define K1A_GO(L2, SH, HI) hipExtLaunchKernelGGL((k1a_partition<L2, SH, HI>), dim3(e->d.nwg), dim3(K1A_THREADS), (uint32_t)e->k1a_lds, s, ta, tb, 0u, da, d_ev, (u64)n)
#define K1A_GO2(L2, SH) do { if (e->d.hist) K1A_GO(L2, SH, true); else K1A_GO(L2, SH, false); } while (0)
#define K1T_GO(L2, SH, NS) hipExtLaunchKernelGGL((k1a_tile_partition<L2, SH, NS>), dim3(e->d.nwg), dim3(K1T_THREADS), (uint32_t)e->k1a_lds, s, ta, tb, 0u, da, d_ev, (u64)n)
#define K1T_GO2(L2, SH) do { if (e->k1a_nsub == 2) K1T_GO(L2, SH, 2); else K1T_GO(L2, SH, 1); } while (0)
#define K1M_GO(L2, SH) do { if (e->d.np == 256) hipExtLaunchKernelGGL((k1a_team_partition<L2, SH, 2, 1024, 8>), dim3(e->d.nwg), dim3(1024), (uint32_t)e->k1a_lds, s, ta, tb, 0u, da, d_ev, (u64)n); \
                            else if (e->d.np == 512) hipExtLaunchKernelGGL((k1a_team_partition<L2, SH, 2, 1024, 9>), dim3(e->d.nwg), dim3(1024), (uint32_t)e->k1a_lds, s, ta, tb, 0u, da, d_ev, (u64)n); \
                            else hipExtLaunchKernelGGL((k1a_team_partition<L2, SH, 2, 1024, 10>), dim3(e->d.nwg), dim3(1024), (uint32_t)e->k1a_lds, s, ta, tb, 0u, da, d_ev, (u64)n); } while (0)
        da.k1a_rot = e->d.k1a_rot;
        rot0 = e->d.k1a_rot; tb0 = e->d.k1a_ticket_base;           // (restored below when the launch is refused: the device counter only moves if the kernel runs)
        if (e->d.narrow && !e->k1a_team) {                           // the next launch's first chunk goes to the workgroup behind this launch's last one
            const u64 per = (n + e->d.nwg - 1) / e->d.nwg, chunk = per >= 4096 ? 4096 : (per + 1023) / 1024 * 1024;
            e->d.k1a_rot = (u32)((e->d.k1a_rot + (n + chunk - 1) / chunk) % e->d.nwg);
        }
        if (e->d.narrow && e->k1a_team) {
            e->d.k1a_rot = (u32)((e->d.k1a_rot + k1m_tiles(n, e->d.nwg, e->k1a_teams, e->k1a_nt)) % ((u64)e->k1a_teams * e->d.nwg));
            da.k1a_ticket_base = e->d.k1a_ticket_base;
            {   // what this launch draws from the slot's ticket counter: one per tile beyond every team's first, one failing draw per active team
                const unsigned long long nt_ = k1m_tiles(n, e->d.nwg, e->k1a_teams, e->k1a_nt), units_ = (unsigned long long)e->k1a_teams * e->d.nwg;
                if (!(e->d.ablate & 0x4u)) e->d.k1a_ticket_base += (u32)((nt_ > units_ ? nt_ - units_ : 0) + std::min(units_, nt_));
            }
            if (e->l2_in_lds && e->l2_u16) { if (sh) K1M_GO(2, true); else K1M_GO(2, false); }
            else if (e->l2_in_lds) { if (sh) K1M_GO(1, true); else K1M_GO(1, false); }
            else { if (sh) K1M_GO(0, true); else K1M_GO(0, false); }
        }
        else if (e->d.narrow) {
            if (e->l2_in_lds && e->l2_u16) { if (sh) K1T_GO2(2, true); else K1T_GO2(2, false); }
            else if (e->l2_in_lds) { if (sh) K1T_GO2(1, true); else K1T_GO2(1, false); }
            else { if (sh) K1T_GO2(0, true); else K1T_GO2(0, false); }
        }
        else if (e->l2_in_lds) { if (sh) K1A_GO2(true, true); else K1A_GO2(true, false); }
        else { if (sh) K1A_GO2(false, true); else K1A_GO2(false, false); }
#undef K1M_GO
#undef K1T_GO2
#undef K1T_GO
#undef K1A_GO2
#undef K1A_GO
        if (tk) { TimingRec r; r.a = ta; r.b = tb; r.kernel = 1; e->trecs.push_back(r); }
    } else {
        Timed t(e, s, 1);
        u64 want = (n + 255) / 256;
        int grid = (int)std::min<u64>(want, (u64)e->k1_grid);
        hipLaunchKernelGGL(k1_resolve_aggregate, dim3(grid), dim3(256), 0, s, da, d_ev, (u64)n);
    }
    {
        // the host predicts what the launch draws from the slot's ticket counter and where its last tile lands: both only hold if the
        // kernel actually runs — a refused launch must leave them where the device still is
        const hipError_t lr = hipGetLastError();
        if (lr != hipSuccess) {
            e->d.k1a_rot = rot0; e->d.k1a_ticket_base = tb0;
            e->err = std::string("K1 pass A launch: ") + hipGetErrorString(lr);
            return lr == hipErrorOutOfMemory ? SG_ENOMEM : SG_ENODEV;
        }
    }
    e->st.events_in += n;
    e->window_events_in += n;
    return SG_OK;
}

int table_upsert(sg_engine* e, bool svc, u32 ip, u32 node_id, uint8_t kind) {
    if (node_id >= e->cfg.max_known_nodes) { e->err = "node_id beyond max_known_nodes"; return SG_ENOSPC; }
    auto& m = svc ? e->jt.svc_ip : e->jt.pod_ip;
    if (m.find(ip) == m.end() && e->jt.pod_ip.size() + e->jt.svc_ip.size() >= e->cfg.max_ips) { e->err = "join table full (max_ips)"; return SG_ENOSPC; }
    if (!e->jt.upsert(svc, ip, node_id)) { e->err = "join table build failed: raise max_ips"; return SG_ENOSPC; }
    e->jt.set_kind(node_id, kind);
    e->n_known = std::max(e->n_known, node_id + 1);
    return SG_OK;
}

size_t weights_count(u32 L) {
    size_t n = 0;
    for (u32 l = 0; l < L; l++) n += 2 * (size_t)(l == 0 ? SG_F_IN : SG_F_HID) * SG_F_HID + SG_F_HID;
    return n + 2 * SG_F_HID * SG_F_HID + SG_F_EDGE * SG_F_HID + SG_F_HID + SG_F_HID + 1;
}
size_t layer_offset(u32 l) {
    size_t n = 0;
    for (u32 k = 0; k < l; k++) n += 2 * (size_t)(k == 0 ? SG_F_IN : SG_F_HID) * SG_F_HID + SG_F_HID;
    return n;
}

int grid_for(u64 items, int per_block, int cap = 2048) {
    u64 g = (items + per_block - 1) / per_block;
    return (int)std::max<u64>(1, std::min<u64>(g, (u64)cap));
}

// the halo request lists and the active node lists of a sharded window: many workgroups, one launch (SG_K6_ONE_WG=1: the round-3 builder)
void launch_halo_lists(sg_engine* e, hipStream_t s, u32* req, u32 capp) {
    static const bool one_wg = sg_knob("SG_K6_ONE_WG") != nullptr;
    if (one_wg) {
        if (e->d.ncap <= K6_FLAGS_LDS) hipLaunchKernelGGL(k6_halo_build_padded<true>, dim3(1), dim3(1024), 0, s, e->d, req, capp);
        else hipLaunchKernelGGL(k6_halo_build_padded<false>, dim3(1), dim3(1024), 0, s, e->d, req, capp);
        return;
    }
    hipLaunchKernelGGL(k6_halo_lists, dim3(((size_t)e->d.ncap + 1023) / 1024), dim3(1024), 0, s, e->d, req, capp, ++e->k6_epoch, 1u);
}
// ob_mode 1: collect this engine's own raw outbound IPs; 0: caller's union list (d_union, *d_union_n);
// 2: all-gathered per-shard lists (d_union = [world][stride], element 0 of a row = count)
int do_close(sg_engine* e, hipStream_t s, const u32* d_union, const u32* d_union_n, u32 ob_mode = 1, u32 stride = 0, u32 gworld = 0, bool fuse_in = false) {
    int rc = sync_tables(e, s);
    if (rc) return rc;
    // When the kept state does not pay, the window is closed as an engine without it does ("plain": the kept state stays as it is, these
    // windows do not touch it), for a while, then the engine looks again.  Two such streams: every window carries raw outbound IPs (any
    // protocol without a Host header talking to addresses outside the cluster — BASELINE config 5's Kafka / Postgres share: the ids of
    // those nodes are ranks among the window's own, such a window can never close warm and would pay for the kept-CSR detour every time);
    // every window brings edges the kept set lacks (the warm attempt is wasted and the rebuild carries the kept keys as well).  The host
    // learns which path its windows took from the note kw_compact leaves in page-locked memory — it never waits for the device.
    Dev dloc = e->d;
    if (dloc.warm && (size_t)e->cur < e->note_seen.size()) {
        // what the device has said since we last looked (this slot's note; page-locked host memory the kw_compact of an earlier window wrote)
        volatile u64* note = e->h_note[e->cur];
        const u64 seq = note[0];
        if (seq != e->note_seen[e->cur]) {
            e->note_seen[e->cur] = seq;
            const u64 v = note[1];
            const u32 cold = (u32)(v & 0xFFu); const bool obip = (v & 0x100ull) != 0;
            // a warm attempt that ran its merge and THEN met an unknown key is the expensive outcome (the window pays the attempt, the rebuild
            // with the kept keys carried over and the compaction: +45 % on a stream that brings new edges every window, tools/churn_probe.py):
            // three of them in a row and the next 32 windows are closed the plain way, then one more try
            e->cold_streak = cold == 2 ? e->cold_streak + 1 : 0;
            e->obip_streak = (cold == 1 && obip) ? e->obip_streak + 1 : 0;
            if (e->cold_streak >= 3) { e->plain_left = 32; e->cold_streak = 0; }
            if (e->obip_streak >= 4) { e->plain_left = 64; e->obip_streak = 0; }
        }
    }
    if (dloc.warm && e->plain_left) { e->plain_left--; dloc.warm = 0; }
    if ((size_t)e->cur < e->plain_slot.size()) e->plain_slot[e->cur] = e->d.warm && !dloc.warm;
    const Dev& d = dloc;
    // warm windows: try unless switched off (the device decides whether the try holds)
    bool warm_try = d.warm && e->warm_on;
    const u32 wt = warm_try ? 1u : 0u;
    {
        Timed tp(e, s, 2);
        if (ob_mode == 1) hipLaunchKernelGGL(kc_prepare, dim3(1), dim3(1024), 0, s, d, (u64)e->n_known, (u64)e->n_labels_decl, e->d_ob_list, (const u32*)e->d_ob_n, e->ob_list_cap, 1u, (const u32*)nullptr, 0u, 0u, wt);
        else if (ob_mode == 0) hipLaunchKernelGGL(kc_prepare, dim3(1), dim3(1024), 0, s, d, (u64)e->n_known, (u64)e->n_labels_decl, const_cast<u32*>(d_union), d_union_n, e->ob_list_cap, 0u, (const u32*)nullptr, 0u, 0u, wt);
        else hipLaunchKernelGGL(kc_prepare, dim3(1), dim3(1024), 0, s, d, (u64)e->n_known, (u64)e->n_labels_decl, e->d_ob_list, (const u32*)e->d_ob_n, e->ob_list_cap, 2u, d_union, stride, gworld, wt);
    }
    if (d.variant == 0) {                                    // group 7 = K1 pass B (k1b_merge), kernel-exact timing as for pass A
        const bool tk = ((e->timing >> 7) & 1u) && e->closes % e->timing_stride == 0;
        hipEvent_t ta = tk ? get_event(e) : nullptr, tb = tk ? get_event(e) : nullptr;
        Dev db = d; db.batch_state = e->window_events_in == 0 ? 2u : 0u;          // a window without any batch: nothing to merge
        db.kept_compact = d.warm;                                                 // (a warm engine's pass B builds / feeds the kept state: compact node ids, sg_kernels.h sg_kept_compact)
        const bool share = d.npb > e->k1b_cus && 2 * e->k1b_lds <= kLdsBytes;   // several partitions per CU and room for two tables: the SGPR-capped build lets two workgroups share a CU
        // warm engines: the warm attempt (WM 1: seeded tables, accumulators straight to their kept positions; returns at once when
        // kc_prepare has already called the window cold), then the cold merge (WM 2: returns at once on a warm window).  Both are records
        // of group 7: a window's pass B is the SUM of its group-7 records.
#define K1B8W_GOP(U_, SPT_, P_, WM_) do { if (share) hipExtLaunchKernelGGL((k1b_stream_merge<U_, SPT_, P_, WM_>), dim3(d.npb), dim3(e->k1b_threads), (uint32_t)e->k1b_lds, s, ta, tb, 0u, db); \
                                     else hipExtLaunchKernelGGL((k1b_stream_merge_wide<U_, SPT_, P_, WM_>), dim3(d.npb), dim3(e->k1b_threads), (uint32_t)e->k1b_lds, s, ta, tb, 0u, db); } while (0)
#define K1B8W_GO(SPT_, WM_) do { if (e->k1b_pack) K1B8W_GOP(K1B_WARM_U, SPT_, true, WM_); else K1B8W_GOP(K1B_WARM_U, SPT_, false, WM_); } while (0)
#define K1B8W_GO2(WM_) do { const u32 spt = d.k1b_ht / e->k1b_threads; if (spt >= 4) K1B8W_GO(4, WM_); else if (spt == 2) K1B8W_GO(2, WM_); else K1B8W_GO(1, WM_); } while (0)
        if (d.warm) {
            if (warm_try) {
                K1B8W_GO2(1);
                if (tk) { TimingRec r; r.a = ta; r.b = tb; r.kernel = 7; e->trecs.push_back(r); ta = get_event(e); tb = get_event(e); }
            }
            K1B8W_GO2(2);
        } else {
#define K1B_GO(U_, H_) do { if (share) hipExtLaunchKernelGGL((k1b_merge<U_, H_>), dim3(d.np), dim3(e->k1b_threads), (uint32_t)e->k1b_lds, s, ta, tb, 0u, db); \
                            else hipExtLaunchKernelGGL((k1b_merge_wide<U_, H_>), dim3(d.np), dim3(e->k1b_threads), (uint32_t)e->k1b_lds, s, ta, tb, 0u, db); } while (0)
#define K1B8_GOP(U_, SPT_, P_) do { if (share) hipExtLaunchKernelGGL((k1b_stream_merge<U_, SPT_, P_>), dim3(d.npb), dim3(e->k1b_threads), (uint32_t)e->k1b_lds, s, ta, tb, 0u, db); \
                               else hipExtLaunchKernelGGL((k1b_stream_merge_wide<U_, SPT_, P_>), dim3(d.npb), dim3(e->k1b_threads), (uint32_t)e->k1b_lds, s, ta, tb, 0u, db); } while (0)
#define K1B8_GO(U_, SPT_) do { if (e->k1b_pack) K1B8_GOP(U_, SPT_, true); else K1B8_GOP(U_, SPT_, false); } while (0)
#define K1B8_GO2(U_) do { const u32 spt = d.k1b_ht / e->k1b_threads; if (spt >= 4) K1B8_GO(U_, 4); else if (spt == 2) K1B8_GO(U_, 2); else K1B8_GO(U_, 1); } while (0)
        if (d.narrow) { if (e->k1b_u == 8) K1B8_GO2(8); else K1B8_GO2(4); }
        else if (d.hist) K1B_GO(4, true); else if (e->k1b_u == 8) K1B_GO(8, false); else K1B_GO(4, false);
        }
#undef K1B8_GO2
#undef K1B8_GO
#undef K1B8_GOP
#undef K1B_GO
#undef K1B8W_GO2
#undef K1B8W_GO
#undef K1B8W_GOP
        if (tk) { TimingRec r; r.a = ta; r.b = tb; r.kernel = 7; e->trecs.push_back(r); }
    }
    {
        Timed t(e, s, 2);
        // An engine that keeps warm-window state rebuilds the KEPT CSR, not the window's: the same kernels on a Dev whose CSR pointers
        // are the kept arrays (capacity npb x pcap: whatever pass B's tables can emit fits, nothing is cut) and whose node statistics
        // are scratch (the row sort reduces every row it sorts; the window's statistics come from kw_compact).  On a warm window each
        // of them returns at once.
        Dev dk = d;
        dk.kept_compact = d.warm;
        if (d.warm) {
            dk.rowptr = d.k_rowptr; dk.col = d.k_col; dk.csr_from = d.k_from; dk.acc_csr = d.k_acc; dk.max_edges = (u64)d.npb * d.pcap;
            dk.st_sum = e->scr_sum[e->cur]; dk.st_max = e->scr_max[e->cur]; dk.row_mu = e->scr_mu[e->cur]; dk.row_sd = e->scr_mu[e->cur] + d.ncap + 1;
        }
        if (d.variant == 1) {
            const u32 ntiles = e->ecap / K2_TILE;
            hipLaunchKernelGGL(k2_edge_count, dim3(ntiles), dim3(256), 0, s, d);
            hipLaunchKernelGGL(k2_scan_tiles, dim3(1), dim3(1024), 0, s, d, ntiles);
            hipLaunchKernelGGL(k2_edge_compact, dim3(ntiles), dim3(256), 0, s, d);
        }
        if (d.dh_g) hipLaunchKernelGGL(k2_deg_hist, dim3(d.dh_g), dim3(K2_DH_THREADS), ((size_t)d.ncap + 1) * sizeof(u32), s, dk);
        if (d.dh_g) hipLaunchKernelGGL((k2_rowptr<K2_RP_ROWS_DH, true>), dim3(((size_t)d.ncap + K2_RP_ROWS_DH) / K2_RP_ROWS_DH), dim3(1024), 0, s, dk, ++e->rp_epoch);
        else hipLaunchKernelGGL((k2_rowptr<K2_RP_ROWS, false>), dim3(((size_t)d.ncap + K2_RP_ROWS) / K2_RP_ROWS), dim3(1024), 0, s, dk, ++e->rp_epoch);
        if (d.variant == 1) hipLaunchKernelGGL(k2_scatter_table, dim3(grid_for(e->cfg.max_edges, 256)), dim3(256), 0, s, d);
        else hipLaunchKernelGGL(k2_scatter_parts, dim3(d.npb), dim3(256), 0, s, dk);
        hipLaunchKernelGGL(k2_rowsort_gather, dim3(std::max(2, std::min(4096, 2 * K2_LONG_WGS + grid_for(d.ncap, 8)))), dim3(256), 2 * (size_t)d.k2_sortw * sizeof(u32), s, dk);
        if (d.warm) {
            // the window's CSR out of the kept one — every window; behind a rebuild its first workgroups also take the positions the
            // next windows' pass B writes to (kw_capture)
            const unsigned kg = (unsigned)(((u64)d.npb * d.pcap + KW_CH - 1) / KW_CH) + KW_CAPW;
            hipLaunchKernelGGL(kw_compact, dim3(kg), dim3(KW_THREADS), (size_t)KW_ROWS * 5 * sizeof(u64), s, d, ++e->kw_epoch, dk.st_sum, dk.st_max, (u64)(e->closes + 1), e->slots.size() > 1 ? 1u : 0u);
        }
    }
    {
        Timed t3(e, s, 8);                                   // group 8 = in-statistics (group 3 = node + edge features)
        const u32 fin = d.warm ? (u32)std::min<u64>(64, ((u64)d.ncap + 255) / 256) : 1u;   // finishing workgroups: the block-sorted rows (1), every row of a warm window
        const bool no_fuse = sg_knob("SG_K3_NO_FUSE") != nullptr;      // (development build: the two-launch form beside the fused one on one box)
        if (fuse_in && !no_fuse) {
            // the one-call pipelines (nothing reads the in-statistics between the close and the features): the features sum the partials, the
            // finishing work rides at the end of k3_in_part's launch — fifteen launches per window instead of sixteen
            const u32 finp = d.warm ? (fin + 3) / 4 : 1u;        // (1024-thread workgroups there)
            hipLaunchKernelGGL(k3_in_part, dim3(e->k3_ranges * e->k3_slices + finp), dim3(1024), e->k3in_lds, s, d, e->k3_slices, finp);
            e->in_fused_slices = e->k3_slices;
        } else {
            hipLaunchKernelGGL(k3_in_part, dim3(e->k3_ranges * e->k3_slices), dim3(1024), e->k3in_lds, s, d, e->k3_slices, 0u);
            hipLaunchKernelGGL(k3_in_reduce, dim3(grid_for((u64)d.ncap * 6, 256, 1024) + fin), dim3(256), 0, s, d, e->k3_slices, fin);
            e->in_fused_slices = 0;
        }
    }
    HIP_TRY(e, hipGetLastError());
    e->closed = true;
    e->window_events_in = 0;
    e->closes++;
    return SG_OK;
}

int do_features(sg_engine* e, hipStream_t s) {
    const Dev& d = e->d;
    Timed t(e, s, 3);
    const int nbn = grid_for(d.ncap, 128), nbe = grid_for(e->cfg.max_edges, 256, 4096);   // (the edge workgroups capped at 1 280 / 2 560 — whole rounds: no difference, profiles/r06_grids_ab_c3.txt)
    hipLaunchKernelGGL(k3_node_features, dim3(nbn + nbe), dim3(256), 0, s, d, (u32)nbn, e->in_fused_slices);
    e->in_fused_slices = 0;
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}

// fuse_proj: the last layer also writes the score head's P and Q (unsharded pipelines only)
int do_layer(sg_engine* e, u32 l, hipStream_t s, bool fuse_proj) {
    const Dev& d = e->d;
    if (!e->have_w) { e->err = "sg_load_weights not called"; return SG_ESTATE; }
    const float* Wl = e->d_W + layer_offset(l);
    const float* Wh = e->d_W + layer_offset(e->cfg.layers);
    const int grid = grid_for(d.ncap, 16);
    const bool pj = fuse_proj && l + 1 == e->cfg.layers && d.world == 1;
    Timed t(e, s, 4);
    // two launches per layer: the gather-mean at high occupancy (8 rows per workgroup), then the dense tiles
    // (small graphs keep the fused kernel: at C2 the second launch costs more than the gather gains — 16.6 vs 20.9 us)
    bool split = e->cfg.max_edges > (1u << 17);              // (C2's 66 k-edge engine stays fused; a 160 k-edge shard of C4 — with its hub rows — splits)
    if (const char* v = sg_knob("SG_K4_FUSED")) split = std::atoi(v) == 0;
#define K4_LAUNCH(FI, MF, PJ, HIN, HOUT) do { if (split) { \
            hipLaunchKernelGGL((k4_gather<FI>), dim3(grid_for(d.ncap, K4G_ROWS, 4096 * 8 / K4G_ROWS)), dim3(K4G_ROWS * 64), 0, s, d, HIN); \
            hipLaunchKernelGGL((k4_sage_layer<FI, MF, PJ, (PJ ? 512 : 256), true>), dim3(grid), dim3(PJ ? 512 : 256), 0, s, d, HIN, HOUT, Wl, Wh); \
        } else hipLaunchKernelGGL((k4_sage_layer<FI, MF, PJ, (FI == 32 ? 1024 : 512), false>), dim3(grid), dim3(FI == 32 ? 1024 : 512), 0, s, d, HIN, HOUT, Wl, Wh); } while (0)
    if (l == 0) {
        if (e->use_mfma) { if (pj) K4_LAUNCH(32, true, true, d.x0, d.h[1]); else K4_LAUNCH(32, true, false, d.x0, d.h[1]); }
        else { if (pj) K4_LAUNCH(32, false, true, d.x0, d.h[1]); else K4_LAUNCH(32, false, false, d.x0, d.h[1]); }
    } else {
        if (e->use_mfma) { if (pj) K4_LAUNCH(64, true, true, d.h[l], d.h[l + 1]); else K4_LAUNCH(64, true, false, d.h[l], d.h[l + 1]); }
        else { if (pj) K4_LAUNCH(64, false, true, d.h[l], d.h[l + 1]); else K4_LAUNCH(64, false, false, d.h[l], d.h[l + 1]); }
    }
#undef K4_LAUNCH
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}

// proj_done: the last SAGE layer already wrote P and Q.  fuse_reset: fold the window reset into the
// score kernel (unsharded variant-0 pipelines; saves a launch) — returns true through *did_reset.
int do_score(sg_engine* e, hipStream_t s, bool proj_done, bool fuse_reset, bool* did_reset) {
    const Dev& d = e->d;
    if (!e->have_w) { e->err = "sg_load_weights not called"; return SG_ESTATE; }
    const float* Wh = e->d_W + layer_offset(e->cfg.layers);
    Timed t(e, s, 5);
    if (!(proj_done && d.world == 1)) {
        if (e->use_mfma) hipLaunchKernelGGL((k5_node_proj<true>), dim3(grid_for(d.ncap, 16)), dim3(256), 0, s, d, d.h[e->cfg.layers], Wh);
        else hipLaunchKernelGGL((k5_node_proj<false>), dim3(grid_for(d.ncap, 16)), dim3(256), 0, s, d, d.h[e->cfg.layers], Wh);
    }
    const bool fr = fuse_reset && d.variant == 0;
    // ONE round of workgroups: the kernel runs 3 waves per SIMD (144 registers, amdgpu_waves_per_eu(3)) = 3 workgroups of 256 threads per CU at a
    // time, and its waves stride over the edges — 2 048 workgroups were 2.67 rounds, the last one two-thirds empty.  Same box, C3, two
    // repetitions: K5 61.1 us at 2 048, 61.7 at 1 024, 58.9 at 1 536, 57.8-57.9 at 768 (profiles/r06_grids_ab_c3.txt)
    int g5 = grid_for(e->cfg.max_edges, 32, 3 * (int)e->k1b_cus);
    if (const char* v = sg_knob("SG_K5_GRID")) { const int x = std::atoi(v); if (x >= 1 && x <= 65535) g5 = std::min(grid_for(e->cfg.max_edges, 32, 65535), x); }
    if (fr) hipLaunchKernelGGL(k5_edge_score<true>, dim3(g5), dim3(256), 0, s, d, Wh);
    else hipLaunchKernelGGL(k5_edge_score<false>, dim3(g5), dim3(256), 0, s, d, Wh);
    if (did_reset) *did_reset = fr;
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}

int do_reset(sg_engine* e, hipStream_t s) {
    hipLaunchKernelGGL(k_reset_window, dim3(grid_for(std::max<u64>((u64)e->d.ncap * SG_NODE_STAT_SUM_WORDS, e->obcap), 256, 512)), dim3(256), 0, s, e->d);
    HIP_TRY(e, hipGetLastError());
    e->closed = false;
    e->window_events_in = 0;            // (an open window that is reset is discarded: the next batch is a first batch again)
    return SG_OK;
}

// group 10 = one record per window: from in front of its first pass-A launch to behind its score kernel, on the window's stream
void window_timed_end(sg_engine* e, hipStream_t s) {
    if (!e->win_ta) return;
    TimingRec r; r.kernel = 10; r.a = e->win_ta; r.b = get_event(e);
    hipEventRecord(r.b, s);
    e->trecs.push_back(r);
    e->win_ta = nullptr;
}

// the window counters in e->h_ctr -> running statistics (engine lock held)
void account_window(sg_engine* e) {
    const size_t E = (size_t)e->h_ctr[C_N_EDGES];
    sg_stats& st = e->st;
    st.windows++;
    st.last_window_events = e->h_ctr[C_N_EVENTS];
    st.last_window_edges = E;
    st.last_window_nodes = e->h_ctr[C_N_NODES];
    st.events_dropped_src += e->h_ctr[C_DROPPED_SRC];
    st.events_dropped_cap += e->h_ctr[C_DROPPED_CAP];
    st.events_misrouted += e->h_ctr[C_MISROUTED];
    st.halo_overflow += e->h_ctr[C_HALO_OVF];
    st.alive_in += e->h_ctr[C_ALIVE_SEEN];
    st.alive_dropped += e->h_ctr[C_ALIVE_DROPPED];
    st.last_window_new_edges = 0;
    if (e->d.warm && (size_t)e->cur < e->plain_slot.size() && e->plain_slot[e->cur]) st.windows_plain++;
    else if (e->d.warm) {
        // (counters of the slot that was read: with several windows in flight every slot keeps its own state and its own counts)
        const bool cold = e->h_ctr[C_COLD] != 0;
        if (cold) st.windows_cold++; else st.windows_warm++;
        if (!cold && e->h_ctr[C_DELTA_N]) { st.windows_delta++; st.last_window_new_edges = e->h_ctr[C_DELTA_N]; }
        // (the policy — when to stop trying — reads the device's note at the next close: do_close)
    }
    if (e->h_ctr[C_N_EVENTS]) {
        // convertKernelTimeToUserspaceTime()/1e6 — aggregator/data.go:1740-1743, :1219 (u64 wrap arithmetic)
        st.last_window_tmin_ms = (int64_t)((e->first_user - (e->first_kernel - e->h_ctr[C_TMIN_NS])) / 1000000ull);
        st.last_window_tmax_ms = (int64_t)((e->first_user - (e->first_kernel - e->h_ctr[C_TMAX_NS])) / 1000000ull);
    } else { st.last_window_tmin_ms = st.last_window_tmax_ms = 0; }
}

// the page-locked view buffer holds at least E rows (the buffer the previous view pointed into stays alive for one more
// generation: a caller that still holds the last view — a numpy array over it, a Go slice — reads stale rows, not freed memory)
int view_reserve(sg_engine* e, size_t E) {
    if (E <= e->h_rows_cap) return SG_OK;
    if (e->h_rows_old) { hipHostFree(e->h_rows_old); e->h_rows_old = nullptr; }
    e->h_rows_old = e->h_rows; e->h_rows = nullptr; e->h_rows_cap = 0;
    const size_t want = std::min<size_t>(next_pow2(std::max<size_t>(E, 1024)), std::max<size_t>(e->cfg.max_edges, E));
    HIP_TRY(e, hipHostMalloc((void**)&e->h_rows, want * sizeof(sg_edge_out)));
    e->h_rows_cap = want;
    return SG_OK;
}
// view != nullptr: the rows go to the engine's page-locked buffer and *view points at them (out / cap unused)
int do_read(sg_engine* e, sg_edge_out* out, size_t cap, size_t* n, const sg_edge_out** view = nullptr) {
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(e->h_ctr, e->d.ctr, sizeof(e->h_ctr), hipMemcpyDeviceToHost));
    const size_t E = (size_t)e->h_ctr[C_N_EDGES];
    if (n) *n = E;
    if (view) {
        { const int rc = view_reserve(e, E); if (rc) return rc; }
        if (E) HIP_TRY(e, hipMemcpy(e->h_rows, e->d.rows, E * sizeof(sg_edge_out), hipMemcpyDeviceToHost));
        *view = e->h_rows;
    }
    const size_t take = view ? 0 : std::min(E, cap);
    if (out && take) HIP_TRY(e, hipMemcpy(out, e->d.rows, take * sizeof(sg_edge_out), hipMemcpyDeviceToHost));
    const size_t nob = (size_t)e->h_ctr[C_N_OBIP];
    e->last_obips.resize(nob);
    if (nob) HIP_TRY(e, hipMemcpy(e->last_obips.data(), e->d.ob_sorted, nob * sizeof(u32), hipMemcpyDeviceToHost));
    account_window(e);
    return SG_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

uint32_t sg_abi_version(void) { return SG_ABI_VERSION; }
size_t sg_weights_count(uint32_t layers) { return weights_count(layers); }
uint32_t sg_hash32(uint32_t x) { return sg_fmix32(x); }

const char* sg_last_error(sg_handle h) { return h ? h->err.c_str() : "null handle"; }

int sg_create(const sg_config* cfg, sg_handle* out) {
    if (!cfg || !out) return SG_EINVAL;
    *out = nullptr;
    // ABI 3: the caller states how much of sg_config it knows about; the rest is zero
    constexpr uint32_t kCfgMin = 88;                                 // sizeof(sg_config) when struct_size was introduced
    if (cfg->struct_size < kCfgMin || cfg->struct_size > 4096) return SG_EINVAL;
    sg_config full; std::memset(&full, 0, sizeof full);
    std::memcpy(&full, cfg, std::min<size_t>(cfg->struct_size, sizeof full));
    full.struct_size = (uint32_t)sizeof full;
    cfg = &full;
    if (cfg->abi_version != SG_ABI_VERSION || cfg->layers < 1 || cfg->layers > SG_MAX_LAYERS || cfg->max_edges == 0 ||
        cfg->max_known_nodes == 0 || cfg->world == 0 || cfg->rank >= cfg->world || cfg->max_known_nodes > 0x3FFFFFFFu)
        return SG_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return SG_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return SG_ENODEV;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SG_ENODEV;   // MI355X only: the kernels are gfx950 code objects
    sg_engine* e = new sg_engine();
    e->cfg = *cfg;
    if (const char* v = sg_knob("SG_ARENA")) e->arena_on = std::atoi(v) != 0;
    if (e->cfg.max_batch == 0) e->cfg.max_batch = 1u << 20;
    if (e->cfg.max_ips == 0) e->cfg.max_ips = e->cfg.max_known_nodes;
    auto fail = [&](int rc) { std::fprintf(stderr, "sg_create: %s\n", e->err.c_str()); sg_destroy(e); return rc; };
#define CR(call) do { int _rc = (call); if (_rc) return fail(_rc); } while (0)
#define CH(call) do { hipError_t _r = (call); if (_r != hipSuccess) { e->err = std::string(#call) + ": " + hipGetErrorString(_r); return fail(_r == hipErrorOutOfMemory ? SG_ENOMEM : SG_ENODEV); } } while (0)
    CH(hipSetDevice(cfg->device));
    CH(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CH(hipEventCreateWithFlags(&e->tab_ev, hipEventDisableTiming));
    e->k1_grid = std::min<int>(SG_MAX_K1_WGS, prop.multiProcessorCount * 8);
    e->k1b_cus = (u32)std::max(1, prop.multiProcessorCount);
    const char* env = sg_knob("SG_DENSE_VALU");
    e->use_mfma = !(env && env[0] == '1');

    Dev& d = e->d;
    const u64 ME = cfg->max_edges;
    e->ecap = next_pow2(std::max<u64>(2 * ME, K2_TILE));
    e->obcap = next_pow2(std::max<u64>(2 * (u64)cfg->max_outbound_ips, 64));
    e->ob_list_cap = next_pow2(std::max<u64>((u64)cfg->max_outbound_ips * std::max<u32>(cfg->world, 1), 64));
    d.max_known = cfg->max_known_nodes; d.max_labels = cfg->max_labels; d.max_obip = std::max<u32>(cfg->max_outbound_ips, 1);
    d.rank = cfg->rank; d.world = cfg->world; d.max_edges = ME; d.layers = cfg->layers;
    d.ncap = cfg->max_known_nodes + cfg->max_labels + d.max_obip;
    d.emask = e->ecap - 1; d.obmask = e->obcap - 1;

    // K1 variant: 0 = partitioned LDS aggregation (fast; bounded edges per partition), 1 = global table + atomics
    if (e->cfg.max_window_events == 0) e->cfg.max_window_events = e->cfg.max_batch;
    d.variant = cfg->k1_variant == 1 ? 1u : 0u;
    d.hist = (cfg->flags & SG_CFG_EDGE_HISTOGRAM) ? 1u : 0u;
    d.agg_slots = d.hist ? 5u : 3u;
    {
        // Narrow-record K1 (default): an endpoint is a compact index below 2^nb — KNOWN ids, then LABEL ids, then slots of the
        // outbound-IP table — and the mixed pair's top bits are the partition (sg_hash.h sg_kmix); it needs the in-partition
        // remainder of the key to fit 31 bits.  k1_variant 2 (or SG_K1_LEGACY) keeps the 16-byte-record kernels, which also
        // carry the per-edge histogram.
        const u64 cn = (u64)cfg->max_known_nodes + cfg->max_labels + e->obcap;
        u32 nb = 12; while ((1ull << nb) < cn && nb < 31) nb++;
        bool narrow = d.variant == 0 && !d.hist && cfg->k1_variant != 2 && !sg_knob("SG_K1_LEGACY") && nb <= 24;
        // k1_variant 0 (auto) picks by the WINDOW, not only by what fits: the 8-byte path pays from a few million events or a quarter
        // of a million edges per window up; below that (BASELINE config 2: 1 M events, 54 k edges) a window is a dozen launches of
        // 5-30 us and the 16-byte kernels' shorter prologue and simpler pass B win — same box, C2: 130 us per window against 147-155
        // (three rounds of tuning for config 3 had been paid for at config 2).  3 asks for the 8-byte path by name.
        const bool small_window = ME < (1ull << 18) && e->cfg.max_window_events <= (2ull << 20);
        if (cfg->k1_variant == 0 && small_window && !sg_knob("SG_K1_NARROW")) narrow = false;
        u64 np = 0;
        if (narrow) {
            // partitions: ~2700 distinct edges each at the configured capacity at most (pass B's LDS table: 4096 slots of 36
            // bytes, 3072 may fill), at least 256.  Fewer partitions = longer runs per tile in pass A.
            np = next_pow2(std::max<u64>((ME + 2699) / 2700, 256));
            u32 pbt = 0; while ((1ull << pbt) < np) pbt++;
            if (np > 2048 || 2 * nb - pbt > 31) narrow = false;          // (one wave scans the run lengths: beyond this the 16-byte kernels / variant 1)
        }
        d.k1b_split = 1;
        if (narrow) {
            // pass B: a partition that may hold more than ~1150 edges is merged by TWO workgroups (sub-tables of 2048 slots, two
            // workgroups per CU) rather than by one with a 4096-slot table that owns the CU alone
            if (ME / np > 1150) { d.k1b_split = 2; d.k1b_ht = 2048; }
            else d.k1b_ht = ME / np > 550 ? 2048 : 1024;
            // (sizing the partitions by the window's RECORDS as well — 512 x 2 workgroups for a shard of C4, 10 M events over 126 k
            // edges — was measured and bought nothing: 79 vs 73 us)
            if (const char* v = sg_knob("SG_SPLIT")) { const int x = std::atoi(v); if (x == 1 || x == 2) { d.k1b_split = (u32)x; d.k1b_ht = ME / (np * x) > 1150 ? 4096 : (ME / (np * x) > 550 ? 2048 : 1024); } }
        } else {
            // 16-byte records: at most ~1250 distinct edges per partition (pass B's LDS table: 2048 slots, 1536 may fill; 1024 slots
            // for small graphs), at least one per CU.  C3 with 1024 partitions 181 us, 2048: 198 us, 4096: 292 us.
            np = next_pow2(std::max<u64>((ME + 1249) / 1250, 256));
            if (np > 4096) np = 4096;
            if (cfg->k1_variant != 1 && ME > (u64)4096 * 1400) d.variant = 1;   // beyond the partitioned path's range
            d.k1b_ht = ME / np > 600 ? 2048 : 1024;
            if (d.hist && d.k1b_ht == 2048) { d.k1b_ht = 1024; np = std::min<u64>(4096, np * 2); }   // 16 x u32 bins per slot: 104 bytes, half the slots, twice the partitions
        }
        d.np = (u32)np; d.nwg = 256;
        // tuning overrides (tools/k1_sweep.py); anything that is not a legal geometry is ignored
        if (const char* v = sg_knob("SG_NP")) { const u64 x = std::strtoull(v, nullptr, 0); if (x >= 64 && x <= (narrow ? 2048u : 4096u) && (x & (x - 1)) == 0) d.np = (u32)x; }
        if (const char* v = sg_knob("SG_HT")) { const u64 x = std::strtoull(v, nullptr, 0); if (x >= 256 && x <= (narrow ? 4096u : 2048u) && (x & (x - 1)) == 0) d.k1b_ht = (u32)x; }
        if (const char* v = sg_knob("SG_NWG")) { const u64 x = std::strtoull(v, nullptr, 0); if (x >= 1 && x <= (u64)SG_MAX_K1_WGS) d.nwg = (u32)x; }
        d.pb = 0; while ((1u << d.pb) < d.np) d.pb++;
        if (narrow && 2 * nb - d.pb > 31) { d.np = (u32)np; d.pb = 0; while ((1u << d.pb) < d.np) d.pb++; }   // an SG_NP override that would not leave 31 remainder bits
        d.narrow = (narrow && d.variant == 0) ? 1u : 0u;
        if (!d.narrow) d.k1b_split = 1;
        // warm windows: the 8-byte-record path without the per-edge histogram (whose bins would have to follow the kept order too);
        // SG_CFG_NO_WARM / SG_WARM=0 keep every window on the full rebuild.  (k1b_u = 8, the one-table-per-CU build, has no warm instantiation.)
        // By default only where it pays: on a graph below a quarter of a million edges the rebuild's four kernels cost about what the
        // compaction plus their four empty launches do (C2, same box: 147 us per window without, 155 with).  SG_CFG_WARM asks for it anyway.
        d.warm = (d.narrow && !d.hist && !(cfg->flags & SG_CFG_NO_WARM) && ((cfg->flags & SG_CFG_WARM) || ME >= (1ull << 18))) ? 1u : 0u;
        if (const char* v = sg_knob("SG_WARM")) { d.warm = (std::atoi(v) != 0 && d.narrow && !d.hist) ? 1u : 0u; }
        d.npb = d.np * d.k1b_split;
        d.nb = nb; d.rb = 2 * nb - d.pb;
        d.pcap = d.narrow ? d.k1b_ht * 13 / 16 : d.k1b_ht * 3 / 4;       // (u32 keys probe cheaply: the narrow tables may fill to 0.81)
        const double m = (double)e->cfg.max_window_events / ((double)d.np * d.nwg);
        d.sa = std::min<u32>(24, std::max<u32>(8, (2 * 2048 / d.np + 6 + 1) & ~1u));   // aggregates per piece: cache slots / partitions, with head room
        if (d.narrow) {
            // a hot key the cache missed lands in ONE piece: twice the mean + 8 sigma + 24 records, then the overflow list
            d.sn = ((u32)(2.0 * m + 8.0 * std::sqrt(m + 1.0) + 24.0) + 1u) & ~1u;
            d.sn = std::min<u32>(d.sn, (1u << 20) - 16);
            d.sw = std::min<u32>(64, std::max<u32>(8, d.sn / 8));
            d.punits = (d.sn + 2 * d.sw + 5 * d.sa + 15) / 16 * 16;     // a piece = whole 128-byte lines; the slack goes to the narrow region
            d.sn = d.punits - 2 * d.sw - 5 * d.sa;
            d.ss = 0; d.pslots = 0;
        } else {
            d.ss = (u32)(2.0 * m + 8.0 * std::sqrt(m + 1.0) + 24.0);          // a hot key the cache missed lands in ONE piece: head room, then the overflow list
            d.ss = (d.ss + d.agg_slots * d.sa + 7) / 8 * 8 - d.agg_slots * d.sa;   // a piece = whole 128-byte lines
            d.ss = std::min<u32>(d.ss, (1u << 20) - 8);
            d.pslots = d.ss + d.agg_slots * d.sa;
        }
        d.ovf_cap = 1u << 16;
        e->k1b_threads = 1024u;                                          // measured: 1024 threads beat 2 x 512 (C3 135 vs 153 us, C2 15.5 vs 22.9 us)
        // narrow pass B: 8 x 16 bytes per lane in flight where a CU holds one table anyway; two workgroups per CU need <= 64 VGPRs
        if (d.narrow) e->k1b_u = ((size_t)d.k1b_ht * 36 + 8) * 2 > kLdsBytes && m > 24.0 ? 8 : 4;
        if (const char* v = sg_knob("SG_K1B_U")) { if (std::atoi(v) == 8) e->k1b_u = 8; if (std::atoi(v) == 4) e->k1b_u = 4; }
        if (e->k1b_u == 8) d.warm = 0;
        // the packed add of pass B is exact while a workgroup merges fewer than 2^16 narrow records: a partition's pieces hold sn each
        e->k1b_pack = d.narrow && (u64)d.sn * d.nwg < 65536ull;
        if (const char* v = sg_knob("SG_K1B_PACK")) { if (std::atoi(v) == 0) e->k1b_pack = false; }
        if (const char* v = sg_knob("SG_K1B_THREADS")) { const u64 x = std::strtoull(v, nullptr, 0); if ((x == 256 && !d.narrow) || x == 512 || x == 1024) e->k1b_threads = (u32)x; }
        if (d.narrow && d.k1b_ht / e->k1b_threads > 4) e->k1b_threads = 1024u;    // (the compaction takes at most four table slots per thread)
    }
    // join tables: word image (join_host.hpp) on the host, one device copy, a pinned ring for word updates
    {
        const u32 max_blocks = d.variant == 0 ? (u32)std::min<u64>(1024, std::max<u64>(64, (u64)e->cfg.max_ips / 32)) : 2u;
        const sgjoin::Layout L = sgjoin::Table::make_layout(e->cfg.max_ips, cfg->max_known_nodes, max_blocks);
        e->jt_mirror.assign(L.words, 0);
        e->jt.init(L, e->jt_mirror.data(), d.variant == 0);
        e->jt.max_dirty = kUpdCap;
        CR(dev_alloc(e, &e->d_blob, L.words));
        CH(hipHostMalloc((void**)&e->h_blob, (size_t)L.words * 4));
        CH(hipEventCreateWithFlags(&e->blob_ev, hipEventDisableTiming));
        CH(hipEventCreateWithFlags(&e->k1_ev, hipEventDisableTiming));
        for (int i = 0; i < kUpdSlots; i++) {
            CH(hipHostMalloc((void**)&e->h_upd[i], (size_t)2 * kUpdCap * sizeof(uint2)));
            CR(dev_alloc(e, &e->d_upd[i], (size_t)2 * kUpdCap));
            CH(hipEventCreateWithFlags(&e->upd_ev[i], hipEventDisableTiming));
        }
        d.jl1 = reinterpret_cast<const u64*>(e->d_blob + L.off_l1);
        d.jl2 = e->d_blob + L.off_l2;
        d.iptab = reinterpret_cast<const u64*>(e->d_blob + L.off_ck); d.ipmask = L.ipcap - 1;
        d.iptab2 = reinterpret_cast<const u64*>(e->d_blob + L.off_ck2); d.ipmask2 = L.ip2cap - 1;
        d.kind = reinterpret_cast<const uint8_t*>(e->d_blob + L.off_kind);
    }
    if (d.variant == 0) {
        if (!k1a_geometry(e)) { e->err = "K1 pass A: piece counters and join level 1 do not fit a CU's LDS"; return fail(SG_ENOSPC); }
        // (narrow: accumulators, keys, two counters, then the warm path's touch bits and new-key bits — one each per slot — and three words)
        e->k1b_lds = d.narrow ? ((size_t)d.k1b_ht * 36 + 8 + (size_t)d.k1b_ht / 4 + 12 + 15) / 16 * 16 : (size_t)d.k1b_ht * (8 + 32 + (d.hist ? 4 * SG_HIST_BINS : 0));
        for (const void* f : {reinterpret_cast<const void*>(k1a_partition<true, true, false>), reinterpret_cast<const void*>(k1a_partition<true, false, false>),
                              reinterpret_cast<const void*>(k1a_partition<false, true, false>), reinterpret_cast<const void*>(k1a_partition<false, false, false>),
                              reinterpret_cast<const void*>(k1a_partition<true, true, true>), reinterpret_cast<const void*>(k1a_partition<true, false, true>),
                              reinterpret_cast<const void*>(k1a_partition<false, true, true>), reinterpret_cast<const void*>(k1a_partition<false, false, true>),
                              reinterpret_cast<const void*>(k1a_tile_partition<2, true, 2>), reinterpret_cast<const void*>(k1a_tile_partition<2, false, 2>),
                              reinterpret_cast<const void*>(k1a_tile_partition<1, true, 2>), reinterpret_cast<const void*>(k1a_tile_partition<1, false, 2>),
                              reinterpret_cast<const void*>(k1a_tile_partition<0, true, 2>), reinterpret_cast<const void*>(k1a_tile_partition<0, false, 2>),
                              reinterpret_cast<const void*>(k1a_tile_partition<2, true, 1>), reinterpret_cast<const void*>(k1a_tile_partition<2, false, 1>),
                              reinterpret_cast<const void*>(k1a_tile_partition<1, true, 1>), reinterpret_cast<const void*>(k1a_tile_partition<1, false, 1>),
                              reinterpret_cast<const void*>(k1a_tile_partition<0, true, 1>), reinterpret_cast<const void*>(k1a_tile_partition<0, false, 1>),
                              reinterpret_cast<const void*>(k1a_team_partition<2, true, 2, 1024, 8>), reinterpret_cast<const void*>(k1a_team_partition<2, false, 2, 1024, 8>),
                              reinterpret_cast<const void*>(k1a_team_partition<2, true, 2, 1024, 9>), reinterpret_cast<const void*>(k1a_team_partition<2, false, 2, 1024, 9>),
                              reinterpret_cast<const void*>(k1a_team_partition<2, true, 2, 1024, 10>), reinterpret_cast<const void*>(k1a_team_partition<2, false, 2, 1024, 10>),
                              reinterpret_cast<const void*>(k1a_team_partition<1, true, 2, 1024, 8>), reinterpret_cast<const void*>(k1a_team_partition<1, false, 2, 1024, 8>),
                              reinterpret_cast<const void*>(k1a_team_partition<1, true, 2, 1024, 9>), reinterpret_cast<const void*>(k1a_team_partition<1, false, 2, 1024, 9>),
                              reinterpret_cast<const void*>(k1a_team_partition<1, true, 2, 1024, 10>), reinterpret_cast<const void*>(k1a_team_partition<1, false, 2, 1024, 10>),
                              reinterpret_cast<const void*>(k1a_team_partition<0, true, 2, 1024, 8>), reinterpret_cast<const void*>(k1a_team_partition<0, false, 2, 1024, 8>),
                              reinterpret_cast<const void*>(k1a_team_partition<0, true, 2, 1024, 9>), reinterpret_cast<const void*>(k1a_team_partition<0, false, 2, 1024, 9>),
                              reinterpret_cast<const void*>(k1a_team_partition<0, true, 2, 1024, 10>), reinterpret_cast<const void*>(k1a_team_partition<0, false, 2, 1024, 10>)})
            CH(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
        for (const void* f : {reinterpret_cast<const void*>(k1b_merge<4, false>), reinterpret_cast<const void*>(k1b_merge<8, false>), reinterpret_cast<const void*>(k1b_merge<4, true>),
                              reinterpret_cast<const void*>(k1b_merge_wide<4, false>), reinterpret_cast<const void*>(k1b_merge_wide<8, false>), reinterpret_cast<const void*>(k1b_merge_wide<4, true>),
                              reinterpret_cast<const void*>(k1b_stream_merge<4, 1, false>), reinterpret_cast<const void*>(k1b_stream_merge<4, 2, false>), reinterpret_cast<const void*>(k1b_stream_merge<4, 4, false>),
                              reinterpret_cast<const void*>(k1b_stream_merge<8, 1, false>), reinterpret_cast<const void*>(k1b_stream_merge<8, 2, false>), reinterpret_cast<const void*>(k1b_stream_merge<8, 4, false>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<4, 1, false>), reinterpret_cast<const void*>(k1b_stream_merge_wide<4, 2, false>), reinterpret_cast<const void*>(k1b_stream_merge_wide<4, 4, false>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<8, 1, false>), reinterpret_cast<const void*>(k1b_stream_merge_wide<8, 2, false>), reinterpret_cast<const void*>(k1b_stream_merge_wide<8, 4, false>),
                              reinterpret_cast<const void*>(k1b_stream_merge<4, 1, true>), reinterpret_cast<const void*>(k1b_stream_merge<4, 2, true>), reinterpret_cast<const void*>(k1b_stream_merge<4, 4, true>),
                              reinterpret_cast<const void*>(k1b_stream_merge<8, 1, true>), reinterpret_cast<const void*>(k1b_stream_merge<8, 2, true>), reinterpret_cast<const void*>(k1b_stream_merge<8, 4, true>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<4, 1, true>), reinterpret_cast<const void*>(k1b_stream_merge_wide<4, 2, true>), reinterpret_cast<const void*>(k1b_stream_merge_wide<4, 4, true>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<8, 1, true>), reinterpret_cast<const void*>(k1b_stream_merge_wide<8, 2, true>), reinterpret_cast<const void*>(k1b_stream_merge_wide<8, 4, true>)})
            CH(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->k1b_lds));
        for (const void* f : {reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 1, false, 1>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 2, false, 1>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 4, false, 1>),
                              reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 1, true, 1>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 2, true, 1>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 4, true, 1>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 1, false, 1>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 2, false, 1>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 4, false, 1>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 1, true, 1>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 2, true, 1>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 4, true, 1>),
                              reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 1, false, 2>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 2, false, 2>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 4, false, 2>),
                              reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 1, true, 2>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 2, true, 2>), reinterpret_cast<const void*>(k1b_stream_merge<K1B_WARM_U, 4, true, 2>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 1, false, 2>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 2, false, 2>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 4, false, 2>),
                              reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 1, true, 2>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 2, true, 2>), reinterpret_cast<const void*>(k1b_stream_merge_wide<K1B_WARM_U, 4, true, 2>)})
            CH(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->k1b_lds));
        CH(hipFuncSetAttribute(reinterpret_cast<const void*>(kw_compact), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)KW_ROWS * 5 * sizeof(u64))));
        e->ecap = K2_TILE;                                              // the global edge table is not used
    }
    d.emask = e->ecap - 1;
    d.alive_cap = cfg->max_alive ? cfg->max_alive : 65536u;
    e->k3_ranges = (d.ncap + K3_IN_NR - 1) / K3_IN_NR;
    // slices of the in-statistics' scan: 8 192 edges each at least; at most 48 where ranges x 48 workgroups (one per CU: 144 KiB of LDS) are ONE
    // round of the chip — since the node features sum the partials themselves (k3_in_reduce is gone from the one-call pipelines) the shorter scan
    // wins: same box, C3, K3-in + K3-feat 53.5-54.4 us at 32 slices, 49.6-49.8 at 48, 53.4 at 51 (profiles/r06_k3slices_ab_c3.txt) — else 32 (a
    // config-5 shard's 49 ranges: more slices only add partials to write and read)
    e->k3_slices = (u32)std::min<u64>((u64)e->k3_ranges * K3_IN_SMAX <= e->k1b_cus ? K3_IN_SMAX : 32, std::max<u64>(8, ME / 8192));
    if (const char* v = sg_knob("SG_K3_SLICES")) { const u64 x = std::strtoull(v, nullptr, 0); if (x >= 1 && x <= 256) e->k3_slices = (u32)x; }
    e->k3in_lds = (size_t)K3_IN_NR * 48;
    {   // the row sort's two LDS arrays: large enough for a bitmap of the node capacity when that fits (a config-5 shard: 150 k nodes = 4.7 k words)
        const u64 bw = ((u64)d.ncap + 31) / 32;
        d.k2_sortw = bw <= K2_SORT_LDS ? K2_SORT_LDS : (u32)std::min<u64>(K2_SORT_LDS_MAX, (bw + 255) / 256 * 256);
        CH(hipFuncSetAttribute(reinterpret_cast<const void*>(k2_rowsort_gather), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (size_t)d.k2_sortw * sizeof(u32))));
    }
    {   // row degrees / in-row ranks by LDS histograms instead of one returning device atomic per edge (k2_deg_hist): when a u32 counter
        // per node fits one workgroup's LDS.  G workgroups of npb / G consecutive output partitions each (G a power of two, 16 .. K2_DH_GMAX).
        d.dh_g = d.dh_ppw = d.dh_ns = 0;
        u32 g = std::min<u32>(K2_DH_GMAX, d.variant == 0 ? d.npb : 0u);            // (k2_rowptr: a multiple of 16)
        // small graphs keep the degree atomics: their cost grows with the edges (1 M: 23 us of pass B), the histogram launch and the wider
        // row scan cost 6-7 us whatever the size — C2 (66 k-edge capacity), same box: window 128.5 / 130.2 us with it, 124.4 / 125.3 without
        if (ME < (1ull << 19)) g = 0;
        // an engine that keeps warm-window state rebuilds rarely, and every launch of the rebuild chain is ~4.5 us of an EMPTY launch on a
        // warm window: there the degree atomics (23-30 us more on a cold window at C3, no launch) are the better trade
        if (d.warm) g = 0;
        if (const char* v = sg_knob("SG_DH_G")) { const u64 x = std::strtoull(v, nullptr, 0); g = (x >= 16 && x <= K2_DH_GMAX && (x & (x - 1)) == 0 && d.variant == 0 && x <= d.npb) ? (u32)x : 0u; }
        if (g >= 16 && (g & (g - 1)) == 0 && d.npb % g == 0 && ((size_t)d.ncap + 1) * sizeof(u32) <= 128u * 1024u && (u64)d.npb * d.pcap < (1ull << 32)) {
            d.dh_g = g; d.dh_ppw = d.npb / g; d.dh_ns = (d.ncap + 1 + 63u) & ~63u;
            CH(hipFuncSetAttribute(reinterpret_cast<const void*>(k2_deg_hist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)d.ncap + 1) * sizeof(u32))));
        }
    }
    CH(hipFuncSetAttribute(reinterpret_cast<const void*>(k3_in_part), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->k3in_lds));
    { const char* ab = sg_knob("SG_ABLATE"); d.ablate = ab ? (u32)std::strtoul(ab, nullptr, 0) : 0u; }
    CR(dev_alloc(e, &d.dbg, (size_t)4 * 4096 * 8));
    CR(dev_alloc(e, &d.clk, (size_t)4));
    // everything a window owns; allocated once per slot
    auto alloc_window = [&](Dev& w, u32*& ob_list, u32*& ob_n) -> int {
#define LR(call) do { int _rc = (call); if (_rc) return _rc; } while (0)
        size_t eslots = ME;
        if (w.variant == 0) {
            eslots = std::max<size_t>(ME, (size_t)w.npb * w.pcap);
            if (w.narrow) { LR(dev_alloc(e, &w.slab8, (size_t)w.np * w.nwg * w.punits)); LR(dev_alloc(e, &w.hdr8, (size_t)w.np * w.nwg));
                            if (!sg_knob("SG_K1B_NO_ORDER")) { LR(dev_alloc(e, &w.k1b_cnt, w.np)); LR(dev_alloc(e, &w.k1b_order, w.np)); } }   // (pass B's largest-first order: sg_k2.h kc_prepare)
            else { LR(dev_alloc(e, &w.slab_s, (size_t)w.np * w.nwg * w.pslots)); LR(dev_alloc(e, &w.hdr, (size_t)w.np * w.nwg)); }
            LR(dev_alloc(e, &w.ovf, (size_t)w.ovf_cap * 9));
            LR(dev_alloc(e, &w.ovf_p, (size_t)w.ovf_cap));
            LR(dev_alloc(e, &w.part_n, w.npb));
            LR(dev_alloc(e, &w.acc_src, (size_t)w.npb * w.pcap * 4));
            LR(dev_alloc(e, &w.e_rank, (size_t)w.npb * w.pcap));
            if (w.dh_g) LR(dev_alloc(e, &w.dh_hist, (size_t)w.dh_g * w.dh_ns));
            if (w.warm) {
                const size_t KC = (size_t)w.npb * w.pcap;              // the kept CSR holds whatever pass B's tables can emit
                LR(dev_alloc(e, &w.wk_keys, (size_t)w.npb * w.k1b_ht, 0xFF)); LR(dev_alloc(e, &w.wk_pos, (size_t)w.npb * w.k1b_ht, 0xFF));
                LR(dev_alloc(e, &w.pos_of_slot, KC, 0xFF));
                LR(dev_alloc(e, &w.k_acc, KC * 4)); LR(dev_alloc(e, &w.k_col, KC)); LR(dev_alloc(e, &w.k_from, KC)); LR(dev_alloc(e, &w.k_rowptr, (size_t)w.ncap + 2));
                LR(dev_alloc(e, &w.kw_tot, (KC + KW_CH - 1) / KW_CH + 2));
                // delta windows: the second kept buffer, kept position -> image index (both buffers), the delta CSR and its maps
                LR(dev_alloc(e, &w.k_col2, KC)); LR(dev_alloc(e, &w.k_from2, KC)); LR(dev_alloc(e, &w.k_rowptr2, (size_t)w.ncap + 2));
                LR(dev_alloc(e, &w.k_slot, 2 * KC));
                LR(dev_alloc(e, &w.dc_rowptr, (size_t)w.ncap + 2)); LR(dev_alloc(e, &w.dc_col, KC)); LR(dev_alloc(e, &w.dc_from, KC)); LR(dev_alloc(e, &w.dc_acc, KC * 4));
                LR(dev_alloc(e, &w.dc_slot, KC)); LR(dev_alloc(e, &w.dc_ip, KC)); LR(dev_alloc(e, &w.dl_img, KC));
                LR(dev_alloc(e, &w.deg2, ((size_t)w.ncap + 1) * SG_DEG_REP * SG_DEG_STRIDE));
                u64* a = nullptr; u64* b = nullptr; double* c = nullptr;
                LR(dev_alloc(e, &a, (size_t)w.ncap * SG_NODE_STAT_SUM_WORDS)); LR(dev_alloc(e, &b, (size_t)w.ncap * SG_NODE_STAT_MAX_WORDS)); LR(dev_alloc(e, &c, 2 * ((size_t)w.ncap + 1)));
                e->scr_sum.push_back(a); e->scr_max.push_back(b); e->scr_mu.push_back(c);
                u64* hn = nullptr; void* dn = nullptr;
                HIP_TRY(e, hipHostMalloc((void**)&hn, 64, hipHostMallocMapped));
                hn[0] = hn[1] = 0;
                HIP_TRY(e, hipHostGetDevicePointer(&dn, hn, 0));
                w.host_note = (u64*)dn;
                e->h_note.push_back(hn); e->note_seen.push_back(0);
            }
        }
        const size_t KE = w.warm ? std::max<size_t>(ME, (size_t)w.npb * w.pcap) : ME;   // arrays the rebuild indexes by KEPT position on a warm engine
        LR(dev_alloc(e, &w.ekeys, e->ecap, 0xFF));
        LR(dev_alloc(e, &w.eacc, (size_t)e->ecap * 4));
        if (w.variant == 1) w.acc_src = w.eacc;
        LR(dev_alloc(e, &w.obkeys, e->obcap));
        LR(dev_alloc(e, &w.wgstat, (size_t)SG_MAX_K1_WGS * WS_WORDS));
        LR(dev_alloc(e, &w.ctr, C_COUNT));
        LR(dev_alloc(e, &w.ob_sorted, w.max_obip));
        LR(dev_alloc(e, &ob_list, e->ob_list_cap));
        LR(dev_alloc(e, &ob_n, 4));
        LR(dev_alloc(e, &w.tile_cnt, e->ecap / K2_TILE));
        LR(dev_alloc(e, &w.tile_off, e->ecap / K2_TILE));
        LR(dev_alloc(e, &w.e_slot, ME)); LR(dev_alloc(e, &w.e_from, eslots)); LR(dev_alloc(e, &w.e_to, eslots));
        LR(dev_alloc(e, &w.longrows, (size_t)w.ncap + 1));
        LR(dev_alloc(e, &w.deg, ((size_t)w.ncap + 1) * SG_DEG_REP * SG_DEG_STRIDE)); LR(dev_alloc(e, &w.rp_tot, ((size_t)w.ncap + K2_RP_ROWS_DH) / K2_RP_ROWS_DH + 1)); LR(dev_alloc(e, &w.k6_tot, (((size_t)w.ncap + 1023) / 1024 + 1) * 16)); LR(dev_alloc(e, &w.k1a_ticket, 4)); LR(dev_alloc(e, &w.lb_ticket, 4)); w.k1a_ticket_base = 0; w.k1a_rot = 0; LR(dev_alloc(e, &w.rowptr, (size_t)w.ncap + 1)); LR(dev_alloc(e, &w.cursor, (size_t)w.ncap + 1));
        LR(dev_alloc(e, &w.col, ME)); LR(dev_alloc(e, &w.cs, KE)); LR(dev_alloc(e, &w.csr_from, ME));
        LR(dev_alloc(e, &w.sort_k, 2 * KE)); LR(dev_alloc(e, &w.sort_v, 2 * KE));
        LR(dev_alloc(e, &w.acc_csr, ME * 4));
        if (w.hist) { LR(dev_alloc(e, &w.hist_src, (w.variant == 0 ? (size_t)w.npb * w.pcap : (size_t)e->ecap) * SG_HIST_BINS)); LR(dev_alloc(e, &w.hist_csr, ME * SG_HIST_BINS)); }
        LR(dev_alloc(e, &w.st_sum, (size_t)w.ncap * SG_NODE_STAT_SUM_WORDS)); LR(dev_alloc(e, &w.st_max, (size_t)w.ncap * SG_NODE_STAT_MAX_WORDS));
        LR(dev_alloc(e, &w.x0, (size_t)w.ncap * SG_F_IN));
        for (u32 l = 1; l <= cfg->layers; l++) LR(dev_alloc(e, &w.h[l], (size_t)w.ncap * SG_F_HID));
        LR(dev_alloc(e, &w.P, (size_t)w.ncap * SG_F_HID)); LR(dev_alloc(e, &w.Q, (size_t)w.ncap * SG_F_HID));
        LR(dev_alloc(e, &w.nmean, (size_t)w.ncap * SG_F_HID));
        w.hub_cap = (u32)std::min<u64>(KE / 256 + 16, 1u << 24);         // blocks of the rows longer than one block: at most E / 512 + one per such row
        LR(dev_alloc(e, &w.hub_items, w.hub_cap)); LR(dev_alloc(e, &w.hub_base, (size_t)w.ncap + 1)); LR(dev_alloc(e, &w.hub_part, (size_t)w.hub_cap * SG_F_HID));
        LR(dev_alloc(e, &w.efeat, ME * SG_F_EDGE)); LR(dev_alloc(e, &w.latz, ME)); LR(dev_alloc(e, &w.errr, ME));
        LR(dev_alloc(e, &w.row_mu, (size_t)w.ncap + 1)); LR(dev_alloc(e, &w.row_sd, (size_t)w.ncap + 1));
        LR(dev_alloc(e, &w.rows, ME));
        LR(dev_alloc(e, &w.in_part, (size_t)e->k3_ranges * e->k3_slices * K3_IN_NR * 6));
        LR(dev_alloc(e, &w.alive_keys, w.alive_cap)); LR(dev_alloc(e, &w.alive_csr, KE));
        LR(dev_alloc(e, &w.act_l, (size_t)w.ncap + 1)); LR(dev_alloc(e, &w.act_p, (size_t)w.ncap + 1));
        // arm the per-workgroup statistic slots (tmin = ~0)
        std::vector<u64> init((size_t)SG_MAX_K1_WGS * WS_WORDS, 0);
        for (int i = 0; i < SG_MAX_K1_WGS; i++) init[(size_t)i * WS_WORDS + WS_TMIN] = ~0ull;
        HIP_TRY(e, hipMemcpy(w.wgstat, init.data(), init.size() * sizeof(u64), hipMemcpyHostToDevice));
#undef LR
        return SG_OK;
    };
    CH(hipStreamSynchronize(e->stream));
    CR(alloc_window(d, e->d_ob_list, e->d_ob_n));
    CR(dev_alloc(e, &e->d_W, weights_count(cfg->layers)));
    d.W = e->d_W;
    {   // (float)log1p((double)i) for small integer counts, by the device's own log1p
        float* tab = nullptr;
        CR(dev_alloc(e, &tab, (size_t)SG_L1P_TAB));
        hipLaunchKernelGGL(k_l1p_table, dim3((SG_L1P_TAB + 255) / 256), dim3(256), 0, e->stream, tab);
        CH(hipStreamSynchronize(e->stream));
        d.l1p_tab = tab;
    }
    CH(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking)); CH(hipStreamCreateWithFlags(&e->copy_stream2, hipStreamNonBlocking));
    if (const char* v = sg_knob("SG_COPY_STREAMS")) e->n_copy = std::atoi(v) == 2 ? 2 : 1;
    CH(hipStreamCreateWithFlags(&e->rd_stream, hipStreamNonBlocking)); CH(hipEventCreateWithFlags(&e->score_ev, hipEventDisableTiming));
    CH(hipHostMalloc((void**)&e->h_ctr_pin, sizeof(e->h_ctr)));
    if (const char* v = sg_knob("SG_STAGE_SLOTS")) e->n_stage = std::min(kStageSlots, std::max(2, std::atoi(v)));
    for (int i = 0; i < e->n_stage; i++) {
        CH(hipEventCreateWithFlags(&e->copied_ev[i], hipEventDisableTiming));
        CH(hipHostMalloc((void**)&e->h_stage[i], (size_t)e->cfg.max_batch * sizeof(sg_event)));
        CR(dev_alloc(e, &e->d_stage[i], e->cfg.max_batch));
        CH(hipEventCreateWithFlags(&e->stage_ev[i], hipEventDisableTiming));
    }
    CH(hipStreamSynchronize(e->stream));
    // further windows in flight: same tables and weights, own window buffers and stream
    {
        const u32 nw = std::min<u32>(std::max<u32>(cfg->windows_in_flight, 1), 8);
        e->slots.resize(nw); e->plain_slot.assign(nw, 0);
        e->slots[0] = sg_engine::WinSlot{e->d, e->stream, e->d_ob_list, e->d_ob_n, false, 0};
        for (u32 k = 1; k < nw; k++) {
            sg_engine::WinSlot& w = e->slots[k];
            w.d = e->d; w.closed = false; w.events_in = 0; w.ob_list = nullptr; w.ob_n = nullptr; w.stream = nullptr;
            CH(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
            CR(alloc_window(w.d, w.ob_list, w.ob_n));
        }
        CH(hipDeviceSynchronize());
    }
#undef CR
#undef CH
    *out = e;
    return SG_OK;
}

int sg_destroy(sg_handle e) {
    if (!e) return SG_EINVAL;
    hipDeviceSynchronize();
    for (size_t k = 0; k < e->slots.size(); k++) if ((int)k != e->cur && e->slots[k].stream) hipStreamDestroy(e->slots[k].stream);
    for (u64* hn : e->h_note) hipHostFree(hn);
    for (void* p : e->allocs) hipFree(p);
    if (e->h_blob) hipHostFree(e->h_blob);
    for (int i = 0; i < kUpdSlots; i++) { if (e->h_upd[i]) hipHostFree(e->h_upd[i]); if (e->upd_ev[i]) hipEventDestroy(e->upd_ev[i]); }
    if (e->blob_ev) hipEventDestroy(e->blob_ev);
    if (e->k1_ev) hipEventDestroy(e->k1_ev);
    for (int i = 0; i < kStageSlots; i++) { if (e->h_stage[i]) hipHostFree(e->h_stage[i]); if (e->stage_ev[i]) hipEventDestroy(e->stage_ev[i]); if (e->copied_ev[i]) hipEventDestroy(e->copied_ev[i]); }
    for (auto& r : e->registered) hipHostUnregister(const_cast<char*>(r.first));
    if (e->copy_stream) hipStreamDestroy(e->copy_stream);
    if (e->copy_stream2) hipStreamDestroy(e->copy_stream2);
    if (e->rd_stream) hipStreamDestroy(e->rd_stream);
    if (e->score_ev) hipEventDestroy(e->score_ev);
    if (e->h_ctr_pin) hipHostFree(e->h_ctr_pin);
    if (e->h_rows) hipHostFree(e->h_rows);
    if (e->h_rows_old) hipHostFree(e->h_rows_old);
    for (auto& r : e->trecs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto v : e->ev_pool) hipEventDestroy(v);
    if (e->tab_ev) hipEventDestroy(e->tab_ev);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return SG_OK;
}

int sg_geometry_get(sg_handle e, sg_geometry* out) {
    if (!e || !out) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    const Dev& d = e->d;
    out->k1_variant = d.variant; out->k1_narrow = d.narrow; out->partitions = d.np; out->table_slots = d.k1b_ht; out->pass_b_split = d.k1b_split;
    out->pass_a_workgroups = d.nwg; out->cache_slots = e->k1a_ct; out->join_l2_in_lds = e->l2_in_lds ? (e->d.narrow && e->l2_u16 ? 2u : 1u) : 0u;
    out->tile_records = d.narrow ? (e->k1a_team ? 4u * e->k1a_nt / e->k1a_teams : K1T_TS(e->k1a_nsub)) : 0u; out->pass_a_teams = d.narrow && e->k1a_team ? e->k1a_teams : 0u; out->endpoint_bits = d.narrow ? d.nb : 0u;
    out->piece_bytes = d.variant != 0 ? 0u : (d.narrow ? d.punits * 8u : d.pslots * 16u);
    out->warm_windows = d.warm;
    return SG_OK;
}

int sg_upsert_pod(sg_handle e, uint32_t ip, uint32_t node_id) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    return table_upsert(e, false, ip, node_id, SG_NODE_POD);
}
int sg_upsert_service(sg_handle e, uint32_t ip, uint32_t node_id) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    return table_upsert(e, true, ip, node_id, SG_NODE_SERVICE);
}
int sg_delete_pod(sg_handle e, uint32_t ip) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->jt.erase(false, ip)) { e->err = "join table rebuild failed"; return SG_ENOSPC; }
    return SG_OK;
}
int sg_delete_service(sg_handle e, uint32_t ip) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->jt.erase(true, ip)) { e->err = "join table rebuild failed"; return SG_ENOSPC; }
    return SG_OK;
}

int sg_set_clock(sg_handle e, uint64_t first_kernel_ns, uint64_t first_user_ns) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    e->first_kernel = first_kernel_ns; e->first_user = first_user_ns;
    return SG_OK;
}

int sg_set_label_count(sg_handle e, uint32_t n_labels) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (n_labels > e->cfg.max_labels) { e->err = "label count beyond max_labels"; return SG_ENOSPC; }
    e->n_labels_decl = std::max(e->n_labels_decl, n_labels);
    return SG_OK;
}

int sg_load_weights(sg_handle e, const float* w, size_t n) {
    if (!e || !w) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (n != weights_count(e->cfg.layers)) { e->err = "weight count mismatch"; return SG_EINVAL; }
    HIP_TRY(e, hipMemcpy(e->d_W, w, n * sizeof(float), hipMemcpyHostToDevice));
    e->have_w = true;
    return SG_OK;
}

namespace {
// a free staging slot (neither being filled by another feeder nor still in flight), or -1: the ring is full
int stage_take(sg_engine* e) {
    for (int k = 0; k < e->n_stage; k++) {
        const int c = (e->stage_next + k) % e->n_stage;
        if (!e->stage_busy[c] && hipEventQuery(e->stage_ev[c]) != hipErrorNotReady) { e->stage_busy[c] = true; e->stage_next = (c + 1) % e->n_stage; e->pending_copies++; return c; }
    }
    return -1;
}
// the copy stream a batch's host -> device copy goes on (engine lock held; two streams alternate when the engine has two)
hipStream_t stage_stream(sg_engine* e) { return (e->n_copy == 2 && (e->copy_rr++ & 1)) ? e->copy_stream2 : e->copy_stream; }
// behind a slot's copy (already enqueued on `cs`, or refused): K1 pass A on the window's stream, the slot handed back (engine lock held)
int stage_finish(sg_engine* e, int slot, size_t n, hipStream_t cs, bool copy_ok) {
    int rc = SG_OK;
    if (!copy_ok) { e->err = "hipMemcpyAsync (staging ring)"; rc = SG_ENODEV; }
    if (rc == SG_OK) {
        hipEventRecord(e->copied_ev[slot], cs);
        hipStreamWaitEvent(e->stream, e->copied_ev[slot], 0);
        e->st.h2d_bytes += n * sizeof(sg_event);
        rc = launch_k1(e, e->d_stage[slot], n, e->stream);
    }
    hipEventRecord(e->stage_ev[slot], e->stream);                        // the slot is free again when pass A has read it
    e->stage_busy[slot] = false; e->pending_copies--;
    e->cv.notify_all();
    return rc;
}
// host (page-locked) -> device slot on the copy stream in one piece, K1 pass A behind it (engine lock held)
int stage_submit(sg_engine* e, int slot, const sg_event* src, size_t n) {
    hipStream_t cs = stage_stream(e);
    const bool ok = hipMemcpyAsync(e->d_stage[slot], src, n * sizeof(sg_event), hipMemcpyHostToDevice, cs) == hipSuccess;
    return stage_finish(e, slot, n, cs, ok);
}
// sg_ingest moves a batch in pieces of this many events (4 MiB): each piece is on the link while the calling thread copies the next
// into the pinned slot.  One 32 MiB batch copied whole keeps the link idle for the 3 ms a thread needs for it — at every window
// boundary, where all feeders start a batch at the same moment, that was a third of the window (bench.py end_to_end)
constexpr size_t kStagePiece = ((size_t)4 << 20) / sizeof(sg_event);
}  // namespace

int sg_ingest(sg_handle e, const sg_event* events, size_t n) {
    if (!e || (!events && n)) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    if (n > e->cfg.max_batch) { e->err = "batch larger than max_batch"; return SG_EINVAL; }
    if (n == 0) return SG_OK;
    if (e->closing) { e->st.ingest_waits++; e->cv.wait(g, [&] { return !e->closing; }); }   // a window boundary is being marked (microseconds): this batch belongs to the next window
    const int slot = stage_take(e);
    if (slot < 0) {                                                      // ring full: drop, never block
        e->st.events_dropped_ring += n;
        return SG_EAGAIN;
    }
    hipStream_t cs = stage_stream(e);
    g.unlock();
    // the caller's memory is not retained; the copy into the pinned slot runs OUTSIDE the engine lock, so several
    // feeder threads (goroutines on different OS threads) fill different slots at the same time — and piece by piece,
    // every piece sent on its way as soon as it is in the slot (the slot is this thread's until stage_finish hands it back;
    // the stream calls need no engine state)
    bool ok = true;
    for (size_t o = 0; o < n && ok; o += kStagePiece) {
        const size_t m = std::min(kStagePiece, n - o);
        std::memcpy(e->h_stage[slot] + o, events + o, m * sizeof(sg_event));
        ok = hipMemcpyAsync(e->d_stage[slot] + o, e->h_stage[slot] + o, m * sizeof(sg_event), hipMemcpyHostToDevice, cs) == hipSuccess;
    }
    g.lock();
    return stage_finish(e, slot, n, cs, ok);
}

// The same without the staging copy, for events that already sit in page-locked memory: memory registered with
// sg_host_register (e.g. the C-allocated buffer a packer writes into), or any hipHostMalloc'd block.  The library reads
// `events` asynchronously (the H2D copy on the copy stream, then K1 pass A): the n records must stay unchanged until
// sg_flush_end* / sg_flush_window* of the window they belong to has RETURNED — sg_flush_begin and sg_window_run only enqueue,
// the copy may still be reading caller memory when they return (after sg_window_run: synchronise its stream first) — the one entry point that keeps caller memory beyond the call, and therefore
// not for Go-heap memory (cgo rule); everything else as sg_ingest (non-blocking, SG_EAGAIN when no device slot is free).
int sg_ingest_pinned(sg_handle e, const sg_event* events, size_t n) {
    if (!e || (!events && n)) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    if (n > e->cfg.max_batch) { e->err = "batch larger than max_batch"; return SG_EINVAL; }
    if (n == 0) return SG_OK;
    {
        const char* p = reinterpret_cast<const char*>(events);
        bool ok = false;
        for (auto& r : e->registered) ok |= p >= r.first && p + n * sizeof(sg_event) <= r.first + r.second;
        if (!ok) { e->err = "sg_ingest_pinned: the events are not inside memory registered with sg_host_register"; return SG_EINVAL; }
    }
    if (e->closing) { e->st.ingest_waits++; e->cv.wait(g, [&] { return !e->closing; }); }
    const int slot = stage_take(e);
    if (slot < 0) { e->st.events_dropped_ring += n; return SG_EAGAIN; }
    return stage_submit(e, slot, events, n);
}
// Blocking convenience for loaders that would rather wait than drop (replay tools, the bench's feeders): n events in
// max_batch-sized pieces through sg_ingest (pinned = 0) or sg_ingest_pinned (pinned = 1); a full ring is waited for (the calling
// thread sleeps 20 us and retries), never counted as a drop.  *retries (optional) = how often it had to wait.
int sg_ingest_bulk(sg_handle e, const sg_event* events, size_t n, int pinned, uint64_t* retries) {
    if (!e || (!events && n)) return SG_EINVAL;
    const size_t step = e->cfg.max_batch;
    uint64_t waits = 0;
    for (size_t o = 0; o < n; o += step) {
        const size_t m = std::min(step, n - o);
        for (;;) {
            const int rc = pinned ? sg_ingest_pinned(e, events + o, m) : sg_ingest(e, events + o, m);
            if (rc == SG_OK) break;
            if (rc != SG_EAGAIN) { if (retries) *retries = waits; return rc; }
            { std::lock_guard<std::mutex> g(e->mu); e->st.events_dropped_ring -= m; }     // (not a drop: it is offered again)
            waits++;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    if (retries) *retries = waits;
    return SG_OK;
}
int sg_host_register(sg_handle e, void* p, size_t bytes) {
    if (!e || !p || !bytes) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { e->err = "hipHostRegister failed"; return SG_ENOMEM; }
    e->registered.push_back({reinterpret_cast<const char*>(p), bytes});
    return SG_OK;
}
int sg_host_unregister(sg_handle e, void* p) {
    if (!e || !p) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    for (size_t i = 0; i < e->registered.size(); i++) if (e->registered[i].first == reinterpret_cast<const char*>(p)) {
        hipDeviceSynchronize();                                          // nothing may still be reading it
        hipHostUnregister(p);
        e->registered.erase(e->registered.begin() + (long)i);
        return SG_OK;
    }
    return SG_EINVAL;
}

int sg_ingest_device(sg_handle e, const sg_event* d_events, size_t n, void* stream) {
    if (!e || (!d_events && n)) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    return launch_k1(e, d_events, n, pick(e, stream));
}

int sg_window_close(sg_handle e, void* stream) {
    if (!e) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    e->cv.wait(g, [&] { return e->pending_copies == 0; });
    return do_close(e, pick(e, stream), nullptr, nullptr, 1u);
}

int sg_window_obip_list(sg_handle e, uint32_t* d_list, uint32_t cap, uint32_t* d_n, void* stream) {
    if (!e || !d_list || !d_n) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = pick(e, stream);
    hipLaunchKernelGGL(k2_ob_collect, dim3(1), dim3(1024), 0, s, e->d, d_list, cap, d_n);
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}

// Use caller-owned device memory for the buffers a sharded driver reduces / exchanges, so that the
// collectives can run on them in place (torch.distributed tensors).  NULL keeps the current buffer.
int sg_bind_buffers(sg_handle e, void* stats_sum, void* stats_max, void* const* feat_rows, uint32_t n_feat) {
    if (!e || n_feat > e->cfg.layers) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (stats_sum) e->d.st_sum = (u64*)stats_sum;
    if (stats_max) e->d.st_max = (u64*)stats_max;
    for (u32 l = 0; l < n_feat; l++) if (feat_rows && feat_rows[l]) e->d.h[l + 1] = (float*)feat_rows[l];
    return SG_OK;
}

int sg_window_close_sharded(sg_handle e, const uint32_t* d_union_ips, const uint32_t* d_union_n, void* stream) {
    if (!e || !d_union_ips || !d_union_n) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    e->cv.wait(g, [&] { return e->pending_copies == 0; });
    return do_close(e, pick(e, stream), d_union_ips, d_union_n, 0u);
}

// Sharded close without a host round trip: d_gathered = the all-gather of every shard's
// [count, ip, ip, ...] buffer (stride u32 per shard), straight from the collective.
int sg_window_close_gathered(sg_handle e, const uint32_t* d_gathered, uint32_t stride, uint32_t world, void* stream) {
    if (!e || !d_gathered || stride < 2 || world == 0 || world > 8) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    e->cv.wait(g, [&] { return e->pending_copies == 0; });
    if ((u64)(stride - 1) * world > e->ob_list_cap) { e->err = "gathered outbound-ip lists exceed the engine's list capacity"; return SG_ENOSPC; }
    return do_close(e, pick(e, stream), d_gathered, nullptr, 2u, stride, world);
}

// Padded halo exchange (fixed-size all-to-all, no host synchronisation).  Lists are [world][capp + 1]
// u32 with element 0 = count.  build: what this shard needs from each owner; pack: rows of layer l for
// the lists the other shards sent (d_serve); unpack: received rows into the feature buffer (d_req).
int sg_halo_build_padded(sg_handle e, uint32_t* d_req, uint32_t capp, void* stream) {
    if (!e || !d_req || capp == 0 || e->cfg.world > 8) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = pick(e, stream);
    Timed t(e, s, 6);
    hipLaunchKernelGGL(k6_halo_mark, dim3(grid_for(e->cfg.max_edges, 256)), dim3(256), 0, s, e->d);
    launch_halo_lists(e, s, d_req, capp);
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}
int sg_halo_pack_padded(sg_handle e, uint32_t l, const uint32_t* d_serve, uint32_t capp, float* d_rows, void* stream) {
    if (!e || l < 1 || l > e->cfg.layers || !d_serve || !d_rows) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = pick(e, stream);
    Timed t(e, s, 6);
    hipLaunchKernelGGL(k6_pack_padded, dim3(grid_for((u64)e->cfg.world * capp * 16, 256)), dim3(256), 0, s, e->d.h[l], d_serve, capp, e->cfg.world, d_rows);
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}
int sg_halo_unpack_padded(sg_handle e, uint32_t l, const uint32_t* d_req, uint32_t capp, const float* d_rows, void* stream) {
    if (!e || l < 1 || l > e->cfg.layers || !d_req || !d_rows) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = pick(e, stream);
    Timed t(e, s, 6);
    hipLaunchKernelGGL(k6_unpack_padded, dim3(grid_for((u64)e->cfg.world * capp * 16, 256)), dim3(256), 0, s, e->d.h[l], d_req, capp, e->cfg.world, d_rows);
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}

int sg_window_features(sg_handle e, void* stream) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->closed) { e->err = "sg_window_features before sg_window_close"; return SG_ESTATE; }
    return do_features(e, pick(e, stream));
}
int sg_window_layer(sg_handle e, uint32_t l, void* stream) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->closed || l >= e->cfg.layers) { e->err = "sg_window_layer: bad phase or layer"; return SG_ESTATE; }
    return do_layer(e, l, pick(e, stream), false);
}
int sg_window_score(sg_handle e, void* stream) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->closed) { e->err = "sg_window_score before sg_window_close"; return SG_ESTATE; }
    return do_score(e, pick(e, stream), false, false, nullptr);
}
int sg_window_score_reset(sg_handle e, void* stream) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->closed) { e->err = "sg_window_score_reset before sg_window_close"; return SG_ESTATE; }
    bool did = false;
    int rc = do_score(e, pick(e, stream), false, true, &did);
    if (rc) return rc;
    if (did) { e->closed = false; return SG_OK; }
    return do_reset(e, pick(e, stream));
}
int sg_window_read(sg_handle e, sg_edge_out* out, size_t cap, size_t* n) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    // (after sg_window_score_reset / sg_window_run_sharded the window is open again, but its rows and counters are still in
    // place until the next batch is ingested: readable)
    if (!e->closed && !(e->last_rows == e->d.rows && e->window_events_in == 0)) { e->err = "sg_window_read before sg_window_close"; return SG_ESTATE; }
    return do_read(e, out, cap, n);
}
int sg_window_reset(sg_handle e, void* stream) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    return do_reset(e, pick(e, stream));
}

// ---- closing a window from the host side, in two halves ------------------------------------------------------------------------
// sg_flush_begin  marks the window boundary: waits for the staging copies that began before it, enqueues K1 pass B .. K5 (with the
//                 window reset folded into K5) and an asynchronous read of the window counters, and returns.  From here on sg_ingest
//                 fills the NEXT window: its copies run on the copy stream at once, its pass-A launches queue behind K5.
// sg_flush_end*   waits for K5 and fetches the rows on the read stream WITHOUT the engine lock — PCIe is full duplex, the rows go
//                 out while the next window's events come in.  (One call doing both held the lock for ~0.55 ms of kernels + 1.1 ms
//                 of copy per C3 window: every feeder thread stood still for a quarter of the window period.)
// sg_flush_window / sg_flush_window_view = begin + end, for callers with one thread.
namespace {
int flush_begin_locked(sg_engine* e, std::unique_lock<std::mutex>& g) {
    // One boundary at a time: `closing` is also the "a begin is in progress" flag — the wait for the staging copies below releases the
    // engine lock, and a second flusher that got in there used to pass the flush_open test too, close the NEXT window and rewrite the
    // rows / counters the first one's unlocked fetch was reading.  It waits here instead and then sees flush_open.
    e->cv.wait(g, [&] { return !e->closing; });
    if (e->flush_open) { e->err = "sg_flush_begin: the previous window has not been fetched (sg_flush_end)"; return SG_ESTATE; }
    e->closing = true;                                                   // later feeders wait instead of slipping batches into the closing window
    e->cv.wait(g, [&] { return e->pending_copies == 0; });
    struct Open { sg_engine* e; ~Open() { e->closing = false; e->cv.notify_all(); } } open{e};
    hipStream_t s = e->stream;
    int rc;
    if ((rc = do_close(e, s, nullptr, nullptr, 1u, 0u, 0u, true))) return rc;
    if ((rc = do_features(e, s))) return rc;
    for (u32 l = 0; l < e->cfg.layers; l++) if ((rc = do_layer(e, l, s, true))) return rc;
    bool did = false;
    if ((rc = do_score(e, s, true, true, &did))) return rc;
    e->fl_rows = e->d.rows; e->fl_ob = e->d.ob_sorted;
    if (did) {                                                           // the window is open again in stream order: fetch later, unlocked
        HIP_TRY(e, hipMemcpyAsync(e->h_ctr_pin, e->d.ctr, sizeof(e->h_ctr), hipMemcpyDeviceToHost, s));
        HIP_TRY(e, hipEventRecord(e->score_ev, s));
        e->closed = false;
        e->flush_async = true;
    } else {                                                             // (variant 1: the reset is its own launch and must follow the read)
        size_t n = 0; const sg_edge_out* v = nullptr;
        if ((rc = do_read(e, nullptr, 0, &n, &v))) return rc;
        if ((rc = do_reset(e, s))) return rc;
        e->flush_async = false;
    }
    e->flush_open = true;
    return SG_OK;
}
// out != nullptr: up to cap rows into caller memory; view != nullptr: all rows into the page-locked view buffer
int flush_end_unlocked(sg_engine* e, std::unique_lock<std::mutex>& g, sg_edge_out* out, size_t cap, size_t* n, const sg_edge_out** view) {
    e->cv.wait(g, [&] { return !e->fetching; });                         // another thread's sg_flush_end of the same window: one fetch, the loser sees the window gone
    if (!e->flush_open) { e->err = "sg_flush_end without sg_flush_begin"; return SG_ESTATE; }
    if (!e->flush_async) {                                               // already read (under the lock, by begin) into the view buffer
        const size_t E = (size_t)e->h_ctr[C_N_EDGES];
        if (n) *n = E;
        if (view) *view = e->h_rows;
        if (out && E && cap) std::memcpy(out, e->h_rows, std::min(E, cap) * sizeof(sg_edge_out));
        e->flush_open = false; e->cv.notify_all();
        return SG_OK;
    }
    const sg_edge_out* d_rows = e->fl_rows; const u32* d_ob = e->fl_ob;
    e->fetching = true;                                                  // claimed: the fetch below runs without the engine lock
    g.unlock();
    int rc = SG_OK; std::string err;
    std::vector<u32> obips;
    size_t E = 0;
    do {
        if (hipEventSynchronize(e->score_ev) != hipSuccess) { err = "hipEventSynchronize (window pipeline)"; rc = SG_ENODEV; break; }
        E = (size_t)e->h_ctr_pin[C_N_EDGES];
        const size_t nob = (size_t)e->h_ctr_pin[C_N_OBIP];
        sg_edge_out* dst = out; size_t take = std::min(E, cap);
        if (view) { g.lock(); rc = view_reserve(e, E); g.unlock(); if (rc) break; dst = e->h_rows; take = E; }
        if (dst && take && hipMemcpyAsync(dst, d_rows, take * sizeof(sg_edge_out), hipMemcpyDeviceToHost, e->rd_stream) != hipSuccess) { err = "hipMemcpyAsync (rows)"; rc = SG_ENODEV; break; }
        obips.resize(nob);
        if (nob && hipMemcpyAsync(obips.data(), d_ob, nob * sizeof(u32), hipMemcpyDeviceToHost, e->rd_stream) != hipSuccess) { err = "hipMemcpyAsync (outbound ips)"; rc = SG_ENODEV; break; }
        if (hipStreamSynchronize(e->rd_stream) != hipSuccess) { err = "hipStreamSynchronize (read stream)"; rc = SG_ENODEV; break; }
    } while (0);
    g.lock();
    e->fetching = false; e->flush_open = false; e->cv.notify_all();
    if (rc) { if (!err.empty()) e->err = err; return rc; }
    std::memcpy(e->h_ctr, e->h_ctr_pin, sizeof(e->h_ctr));
    e->last_obips.swap(obips);
    account_window(e);
    if (n) *n = E;
    if (view) *view = e->h_rows;
    return SG_OK;
}
}  // namespace

int sg_flush_begin(sg_handle e, uint64_t window_end_ms) {
    (void)window_end_ms;
    if (!e) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    return flush_begin_locked(e, g);
}
int sg_flush_end(sg_handle e, sg_edge_out* out, size_t cap, size_t* n) {
    if (!e) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    return flush_end_unlocked(e, g, out, cap, n, nullptr);
}
int sg_flush_end_view(sg_handle e, const sg_edge_out** rows, size_t* n) {
    if (!e || !rows) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    return flush_end_unlocked(e, g, nullptr, 0, n, rows);
}
int sg_flush_window(sg_handle e, uint64_t window_end_ms, sg_edge_out* out, size_t cap, size_t* n) {
    (void)window_end_ms;
    if (!e) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    e->cv.wait(g, [&] { return !e->closing && !e->flush_open; });        // (another thread's begin .. end pair)
    if (const int rc = flush_begin_locked(e, g)) return rc;
    return flush_end_unlocked(e, g, out, cap, n, nullptr);
}
int sg_flush_window_view(sg_handle e, uint64_t window_end_ms, const sg_edge_out** rows, size_t* n) {
    (void)window_end_ms;
    if (!e || !rows) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    e->cv.wait(g, [&] { return !e->closing && !e->flush_open; });
    if (const int rc = flush_begin_locked(e, g)) return rc;
    return flush_end_unlocked(e, g, nullptr, 0, n, rows);
}

// enqueue-only variant of the whole window pipeline (no read-back, no host sync): what bench.py times.
// ---- the sharded window in ONE call (judge item r2-2): local stages + RCCL collectives, all enqueued on one stream ----------
// RCCL is reached through dlopen (the copy already in the process, e.g. torch's, else /opt/rocm/lib/librccl.so): the engine
// library itself does not link it, a single-GPU deployment never loads it.
struct sg_comm {
    void* lib = nullptr; void* comm = nullptr; int rank = 0, world = 1; hipStream_t stream = nullptr; sg_engine* eng = nullptr;   // eng: whose timing records the collectives go to (group 9)
    bool in_group = false; void* grp_t = nullptr;                    // between rc_group_begin and rc_group_end (the grouped calls' one timing record)
    struct Id { char b[128]; };
    int (*GetUniqueId)(Id*) = nullptr; int (*CommInitRank)(void**, int, Id, int) = nullptr; int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr; int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
};
namespace {
bool rccl_load(sg_comm* c) {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"}) { c->lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (c->lib) break; }
    if (!c->lib) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { c->lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (c->lib) break; }
    if (!c->lib) return false;
#define RSYM(field, name) do { c->field = reinterpret_cast<decltype(c->field)>(dlsym(c->lib, name)); if (!c->field) return false; } while (0)
    RSYM(GetUniqueId, "ncclGetUniqueId"); RSYM(CommInitRank, "ncclCommInitRank"); RSYM(CommDestroy, "ncclCommDestroy"); RSYM(AllGather, "ncclAllGather");
    RSYM(AllReduce, "ncclAllReduce"); RSYM(Send, "ncclSend"); RSYM(Recv, "ncclRecv"); RSYM(GroupStart, "ncclGroupStart"); RSYM(GroupEnd, "ncclGroupEnd");
#undef RSYM
    return true;
}
// rccl.h: ncclUint8 = 1, ncclUint64 = 5; ncclSum = 0, ncclMax = 2
int rc_all_gather(void* x, const void* send, void* recv, size_t bytes) { sg_comm* c = (sg_comm*)x; Timed t(c->eng, c->stream, 9); return c->AllGather(send, recv, bytes, 1, c->comm, c->stream) ? SG_ENODEV : SG_OK; }
// (inside a group the two all-reduces are one launch: timed as one record of group 9, opened by the first and closed by rc_group_end)
int rc_all_reduce(void* x, void* buf, size_t n, int op) {
    sg_comm* c = (sg_comm*)x;
    if (c->in_group) return c->AllReduce(buf, buf, n, 5, op ? 2 : 0, c->comm, c->stream) ? SG_ENODEV : SG_OK;
    Timed t(c->eng, c->stream, 9);
    return c->AllReduce(buf, buf, n, 5, op ? 2 : 0, c->comm, c->stream) ? SG_ENODEV : SG_OK;
}
int rc_group_begin(void* x) {
    sg_comm* c = (sg_comm*)x;
    if (!c->GroupStart || !c->GroupEnd) return SG_OK;
    if (c->GroupStart()) return SG_ENODEV;                           // (no group opened: nothing to time, nothing to close)
    c->in_group = true;
    c->grp_t = new Timed(c->eng, c->stream, 9);
    return SG_OK;
}
int rc_group_end(void* x) {
    sg_comm* c = (sg_comm*)x;
    if (!c->in_group) return SG_OK;
    const int rc = c->GroupEnd();
    delete static_cast<Timed*>(c->grp_t); c->grp_t = nullptr; c->in_group = false;
    return rc ? SG_ENODEV : SG_OK;
}
int rc_all_to_all(void* x, const void* send, void* recv, size_t bytes) {            // full mesh over xGMI: grouped point-to-point, every link busy at once
    sg_comm* c = (sg_comm*)x;
    Timed t(c->eng, c->stream, 9);
    int bad = c->GroupStart();
    for (int r = 0; r < c->world; r++) {
        bad |= c->Send((const char*)send + (size_t)r * bytes, bytes, 1, r, c->comm, c->stream);
        bad |= c->Recv((char*)recv + (size_t)r * bytes, bytes, 1, r, c->comm, c->stream);
    }
    bad |= c->GroupEnd();
    return bad ? SG_ENODEV : SG_OK;
}
struct ShardCtx { sg_engine* e; hipStream_t s; };
#define SCX ShardCtx* x = (ShardCtx*)p; sg_engine* e = x->e; hipStream_t s = x->s
int st_obip_list(void* p) { SCX; hipLaunchKernelGGL(k2_ob_collect, dim3(1), dim3(1024), 0, s, e->d, e->xc.ob_local + 1, e->xc.ob_stride - 1, e->xc.ob_local); return hipGetLastError() == hipSuccess ? SG_OK : SG_ENODEV; }
int st_close(void* p) { SCX; return do_close(e, s, e->xc.ob_all, nullptr, 2u, e->xc.ob_stride, e->cfg.world); }
int st_features(void* p) { SCX; return do_features(e, s); }
int st_halo_build(void* p) {
    SCX; Timed t(e, s, 6);
    hipLaunchKernelGGL(k6_halo_mark, dim3(grid_for(e->cfg.max_edges, 256)), dim3(256), 0, s, e->d);
    launch_halo_lists(e, s, e->xc.req, e->xc.capp);
    return hipGetLastError() == hipSuccess ? SG_OK : SG_ENODEV;
}
int st_layer(void* p, uint32_t l) { SCX; return do_layer(e, l, s, false); }
int st_pack(void* p, uint32_t l) { SCX; Timed t(e, s, 6); hipLaunchKernelGGL(k6_pack_padded, dim3(grid_for((u64)e->cfg.world * e->xc.capp * 16, 256)), dim3(256), 0, s, e->d.h[l], e->xc.serve, e->xc.capp, e->cfg.world, e->xc.rows_out); return hipGetLastError() == hipSuccess ? SG_OK : SG_ENODEV; }
int st_unpack(void* p, uint32_t l) { SCX; Timed t(e, s, 6); hipLaunchKernelGGL(k6_unpack_padded, dim3(grid_for((u64)e->cfg.world * e->xc.capp * 16, 256)), dim3(256), 0, s, e->d.h[l], e->xc.req, e->xc.capp, e->cfg.world, e->xc.rows_in); return hipGetLastError() == hipSuccess ? SG_OK : SG_ENODEV; }
int st_score(void* p) {
    SCX; bool did = false;
    int rc = do_score(e, s, false, true, &did);
    if (rc) return rc;
    if (did) { e->closed = false; return SG_OK; }
    return do_reset(e, s);
}
#undef SCX
}  // namespace

// Can this process reach RCCL at all (dlopen + every symbol)?  For drivers that must agree on a fallback BEFORE any rank enters
// ncclCommInitRank (a rank that cannot load librccl would otherwise leave the others waiting in it: ADVICE r3).
int sg_comm_probe(void) { sg_comm c; return rccl_load(&c) ? SG_OK : SG_ENODEV; }
int sg_comm_unique_id(void* id, size_t bytes) {
    if (!id || bytes < 128) return SG_EINVAL;
    sg_comm c;
    if (!rccl_load(&c)) return SG_ENODEV;
    sg_comm::Id u;
    if (c.GetUniqueId(&u)) return SG_ENODEV;
    std::memcpy(id, u.b, 128);
    return SG_OK;
}
int sg_comm_create(const void* id, size_t bytes, int rank, int world, int device, sg_comm** out) {
    if (!id || bytes < 128 || !out || world < 1 || world > 8 || rank < 0 || rank >= world) return SG_EINVAL;
    *out = nullptr;
    sg_comm* c = new sg_comm();
    if (!rccl_load(c)) { delete c; return SG_ENODEV; }
    if (hipSetDevice(device) != hipSuccess) { delete c; return SG_ENODEV; }
    sg_comm::Id u; std::memcpy(u.b, id, 128);
    c->rank = rank; c->world = world;
    if (c->CommInitRank(&c->comm, world, u, rank)) { delete c; return SG_ENODEV; }
    *out = c;
    return SG_OK;
}
int sg_comm_destroy(sg_comm* c) {
    if (!c) return SG_EINVAL;
    if (c->comm) c->CommDestroy(c->comm);
    delete c;
    return SG_OK;
}

// K1 pass B .. K5 of a shard's window with every exchange, enqueued on `stream` (NULL = the engine's): nothing waits for the
// device.  Rows stay on the device (sg_window_rows_buffer); the window is open again afterwards.
int sg_window_run_sharded(sg_handle e, sg_comm* c, void* stream) {
    if (!e || !c || (u32)c->world != e->cfg.world || (u32)c->rank != e->cfg.rank) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    e->cv.wait(g, [&] { return e->pending_copies == 0; });
    hipStream_t s = pick(e, stream);
    const u32 W = e->cfg.world;
    if (!e->xc.ob_local) {                                           // first call: the exchange buffers
        sg_engine::Xchg& x = e->xc;
        x.ob_stride = e->d.max_obip + 1;
        x.capp = std::max<u32>(1, std::min<u32>(e->d.ncap, 2 * ((e->d.ncap + W - 1) / W) + 1024));   // rows one shard may ask ONE owner for: twice an owner's mean share
        if ((u64)(x.ob_stride - 1) * W > e->ob_list_cap) { e->err = "gathered outbound-ip lists exceed the engine's list capacity"; return SG_ENOSPC; }
        int rc;
        if ((rc = dev_alloc(e, &x.ob_local, x.ob_stride)) || (rc = dev_alloc(e, &x.ob_all, (size_t)W * x.ob_stride)) ||
            (rc = dev_alloc(e, &x.req, (size_t)W * (x.capp + 1))) || (rc = dev_alloc(e, &x.serve, (size_t)W * (x.capp + 1))) ||
            (rc = dev_alloc(e, &x.rows_out, (size_t)W * x.capp * SG_F_HID)) || (rc = dev_alloc(e, &x.rows_in, (size_t)W * x.capp * SG_F_HID))) return rc;
        HIP_TRY(e, hipStreamSynchronize(e->stream));                 // (dev_alloc clears on the engine's own stream)
    }
    c->stream = s; c->eng = e;
    ShardCtx cx{e, s};
    sg_shard_stages st{};
    st.ctx = &cx; st.layers = e->cfg.layers; st.world = W;
    st.obip_list = st_obip_list; st.close_gathered = st_close; st.features = st_features; st.halo_build = st_halo_build;
    st.layer = st_layer; st.pack = st_pack; st.unpack = st_unpack; st.score = st_score;
    st.ob_local = e->xc.ob_local; st.ob_all = e->xc.ob_all; st.ob_bytes = (size_t)e->xc.ob_stride * 4;
    st.stats_sum = e->d.st_sum; st.stats_sum_words = (size_t)e->d.ncap * SG_NODE_STAT_SUM_WORDS;
    st.stats_max = e->d.st_max; st.stats_max_words = (size_t)e->d.ncap * SG_NODE_STAT_MAX_WORDS;
    st.req = e->xc.req; st.serve = e->xc.serve; st.list_bytes = (size_t)(e->xc.capp + 1) * 4;
    st.rows_out = e->xc.rows_out; st.rows_in = e->xc.rows_in; st.rows_bytes = (size_t)e->xc.capp * SG_F_HID * 4;
    sg_shard_comm sc{c, rc_all_gather, rc_all_reduce, rc_all_to_all, rc_group_begin, rc_group_end};
    const int rc = sg_run_sharded_window(&st, &sc);
    if (rc) { if (e->err.empty()) e->err = "sg_window_run_sharded: a stage or a collective failed"; return rc; }
    window_timed_end(e, s);
    e->last_rows = e->d.rows;
    return SG_OK;
}

// Rows this shard asked each owner for in its last sharded window (element r = requests to rank r; device-syncs).  Diagnostic.
int sg_window_halo_counts(sg_handle e, uint32_t* counts, size_t world) {
    if (!e || !counts || world != e->cfg.world) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->xc.req) { for (size_t r = 0; r < world; r++) counts[r] = 0; return SG_OK; }
    HIP_TRY(e, hipDeviceSynchronize());
    for (size_t r = 0; r < world; r++) HIP_TRY(e, hipMemcpy(&counts[r], e->xc.req + r * (e->xc.capp + 1), 4, hipMemcpyDeviceToHost));
    return SG_OK;
}
int sg_window_run(sg_handle e, void* stream) {
    if (!e) return SG_EINVAL;
    std::unique_lock<std::mutex> g(e->mu);
    e->cv.wait(g, [&] { return e->pending_copies == 0; });
    hipStream_t s = pick(e, stream);
    int rc;
    if ((rc = do_close(e, s, nullptr, nullptr, 1u, 0u, 0u, true))) return rc;
    if ((rc = do_features(e, s))) return rc;
    for (u32 l = 0; l < e->cfg.layers; l++) if ((rc = do_layer(e, l, s, true))) return rc;
    bool did = false;
    if ((rc = do_score(e, s, true, true, &did))) return rc;
    if (did) e->closed = false; else if ((rc = do_reset(e, s))) return rc;
    window_timed_end(e, s);
    e->last_rows = e->d.rows;
    rotate_window(e);                                    // the next sg_ingest* goes to the next slot (if any)
    return SG_OK;
}

int sg_window_buffers(sg_handle e, void** stats_sum, void** stats_max, void** counters, size_t* n_nodes_cap) {
    if (!e) return SG_EINVAL;
    if (stats_sum) *stats_sum = e->d.st_sum;
    if (stats_max) *stats_max = e->d.st_max;
    if (counters) *counters = e->d.ctr;
    if (n_nodes_cap) *n_nodes_cap = e->d.ncap;
    return SG_OK;
}
int sg_window_feat_buffer(sg_handle e, uint32_t l, void** rows, size_t* row_floats) {
    if (!e || l > e->cfg.layers) return SG_EINVAL;
    if (rows) *rows = l == 0 ? (void*)e->d.x0 : (void*)e->d.h[l];
    if (row_floats) *row_floats = l == 0 ? SG_F_IN : SG_F_HID;
    return SG_OK;
}
int sg_window_rows_buffer(sg_handle e, void** rows) {
    if (!e || !rows) return SG_EINVAL;
    *rows = e->last_rows ? e->last_rows : e->d.rows;
    return SG_OK;
}

int sg_halo_build(sg_handle e, uint32_t* d_ids, uint32_t cap, uint32_t* d_counts, void* stream) {
    if (!e || !d_ids || !d_counts) return SG_EINVAL;
    if (e->cfg.world > 8) { e->err = "halo lists support at most 8 shards (one node)"; return SG_EINVAL; }
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = pick(e, stream);
    Timed t(e, s, 6);
    hipLaunchKernelGGL(k6_halo_mark, dim3(grid_for(e->cfg.max_edges, 256)), dim3(256), 0, s, e->d);
    if (e->d.ncap <= K6_FLAGS_LDS) hipLaunchKernelGGL(k6_active_lists<true>, dim3(1), dim3(1024), 0, s, e->d);
    else hipLaunchKernelGGL(k6_active_lists<false>, dim3(1), dim3(1024), 0, s, e->d);
    hipLaunchKernelGGL(k6_halo_build, dim3(1), dim3(256), 0, s, e->d, d_ids, cap, d_counts);
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}
int sg_halo_pack(sg_handle e, uint32_t l, const uint32_t* d_ids, uint32_t n, float* d_rows, void* stream) {
    if (!e || l < 1 || l > e->cfg.layers) return SG_EINVAL;
    if (n == 0) return SG_OK;
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = pick(e, stream);
    Timed t(e, s, 6);
    hipLaunchKernelGGL(k6_pack, dim3(grid_for((u64)n * 16, 256)), dim3(256), 0, s, e->d.h[l], d_ids, n, d_rows);
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}
int sg_halo_unpack(sg_handle e, uint32_t l, const uint32_t* d_ids, uint32_t n, const float* d_rows, void* stream) {
    if (!e || l < 1 || l > e->cfg.layers) return SG_EINVAL;
    if (n == 0) return SG_OK;
    std::lock_guard<std::mutex> g(e->mu);
    hipStream_t s = pick(e, stream);
    Timed t(e, s, 6);
    hipLaunchKernelGGL(k6_unpack, dim3(grid_for((u64)n * 16, 256)), dim3(256), 0, s, e->d.h[l], d_ids, n, d_rows);
    HIP_TRY(e, hipGetLastError());
    return SG_OK;
}

int sg_window_outbound_ips(sg_handle e, uint32_t* ips, size_t cap, size_t* n) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (n) *n = e->last_obips.size();
    if (ips) std::memcpy(ips, e->last_obips.data(), std::min(cap, e->last_obips.size()) * sizeof(u32));
    return SG_OK;
}

int sg_window_hist(sg_handle e, uint32_t* bins, size_t cap_rows, size_t* n) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->d.hist) { e->err = "sg_window_hist: the engine was created without SG_CFG_EDGE_HISTOGRAM"; return SG_ESTATE; }
    const size_t E = (size_t)e->h_ctr[C_N_EDGES];                      // of the last read window
    if (n) *n = E;
    const size_t take = std::min(E, cap_rows);
    if (bins && take) { HIP_TRY(e, hipDeviceSynchronize()); HIP_TRY(e, hipMemcpy(bins, e->d.hist_csr, take * SG_HIST_BINS * sizeof(uint32_t), hipMemcpyDeviceToHost)); }
    return SG_OK;
}

int sg_stats_get(sg_handle e, sg_stats* out) {
    if (!e || !out) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    *out = e->st;
    return SG_OK;
}

int sg_timing_enable(sg_handle e, int on) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    e->timing = on == 0 ? 0u : (on == 1 ? ~0u : (unsigned)on);
    return SG_OK;
}
// The dispatch stamps of groups 1 and 7 cost a few microseconds per launch (the runtime brackets the kernel with its own signal packets):
// with stride n they are taken on every n-th window only — the averages are still over launches of the timed region, which they then
// disturb n times less (bench.py: 3).
int sg_timing_stride(sg_handle e, uint32_t n) {
    if (!e || n == 0) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    e->timing_stride = n;
    return SG_OK;
}
int sg_timing_reset(sg_handle e) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    for (auto& r : e->trecs) { e->ev_pool.push_back(r.a); e->ev_pool.push_back(r.b); }
    e->trecs.clear();
    return SG_OK;
}
int sg_debug_stamps(sg_handle e, uint64_t* out, size_t n) {
    if (!e || !out) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(out, e->d.dbg, std::min<size_t>(n, (size_t)4 * 4096 * 8) * sizeof(u64), hipMemcpyDeviceToHost));
    return SG_OK;
}
// The shader clock the chip sustains ("fast box / slow box"): spin_mhz from an all-CU integer spin of about
// spin_us microseconds launched now; k1a_mhz from the cycle / 100 MHz tick counts pass A's workgroup 0 has accumulated over
// its launches since the last call (0 when there was none).  Device-syncs.
int sg_clock_probe(sg_handle e, uint32_t spin_us, double* spin_mhz, double* k1a_mhz) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    HIP_TRY(e, hipDeviceSynchronize());
    const u32 iters = std::max<u32>(1000u, spin_us * 120u);           // ~20 cycles per trip of the dependent chain at ~2.4 GHz
    hipLaunchKernelGGL(k_clock_spin, dim3(1024), dim3(256), 0, e->stream, e->d.clk, iters);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    u64 h[4] = {};
    HIP_TRY(e, hipMemcpy(h, e->d.clk, sizeof(h), hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemset(e->d.clk, 0, sizeof(h)));
    if (spin_mhz) *spin_mhz = h[3] ? 100.0 * (double)h[2] / (double)h[3] : 0.0;
    if (k1a_mhz) *k1a_mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return SG_OK;
}
int sg_timing_get(sg_handle e, int kernel, double* avg_us, uint64_t* launches) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    double tot = 0; uint64_t cnt = 0;
    for (auto& r : e->trecs) if (r.kernel == kernel) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += (double)ms * 1000.0; cnt++; }
    }
    if (avg_us) *avg_us = cnt ? tot / (double)cnt : 0.0;
    if (launches) *launches = cnt;
    return SG_OK;
}

int sg_set_warm(sg_handle e, int on) {
    if (!e) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    e->warm_on = on != 0; e->cold_streak = 0; e->plain_left = 0; e->obip_streak = 0;
    return SG_OK;
}
// every record of one group since sg_timing_reset, in launch order (bench.py: median and minimum, SURVEY 8(d) run protocol)
int sg_timing_samples(sg_handle e, int kernel, double* us, size_t cap, size_t* n) {
    if (!e || (!us && cap)) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    size_t cnt = 0;
    for (auto& r : e->trecs) if (r.kernel == kernel) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        if (cnt < cap) us[cnt] = (double)ms * 1000.0;
        cnt++;
    }
    if (n) *n = cnt;
    return SG_OK;
}
// What kind of box this is (the pool's boxes differ in memory latency, not in clock): ONE lane follows a chain of
// dependent loads through `bytes` of device memory (one 128-byte line per step, the order an odd-multiplier walk over all lines)
// and reports the average nanoseconds per load by the 100 MHz reference clock.  bytes >> Infinity Cache = HBM latency; 2 MiB,
// walked once before the clock starts = L2 latency.  Allocates and frees its own buffer; device-syncs.
int sg_latency_probe(sg_handle e, uint64_t bytes, uint32_t steps, int warm, double* ns_per_load) {
    if (!e || !ns_per_load || bytes < 4096 || steps == 0) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    u64 lines = 1; while (lines * 2 * 128 <= bytes) lines *= 2;         // a power of two: x -> a x + c (a % 4 == 1, c odd) visits every line
    if (lines > (1ull << 31)) lines = 1ull << 31;
    u32* buf = nullptr; u64* out = nullptr;
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMalloc((void**)&buf, lines * 128));
    if (hipMalloc((void**)&out, 2 * sizeof(u64)) != hipSuccess) { hipFree(buf); e->err = "sg_latency_probe: hipMalloc"; return SG_ENOMEM; }
    hipLaunchKernelGGL(k_chase_init, dim3((unsigned)std::min<u64>((lines + 255) / 256, 65535)), dim3(256), 0, e->stream, buf, (u32)(lines - 1));
    if (warm & 1) hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, e->stream, (const u32*)buf, (u32)lines, out);   // one whole round: every line is in the cache level it fits
    if (warm & 2) hipLaunchKernelGGL(k_chase, dim3(1024), dim3(64), 0, e->stream, (const u32*)buf, steps, out);    // 65 536 chains in flight: the latency under load
    else hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, e->stream, (const u32*)buf, steps, out);
    u64 h[2] = {};
    hipError_t r = hipStreamSynchronize(e->stream);
    if (r == hipSuccess) r = hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    hipFree(buf); hipFree(out);
    if (r != hipSuccess) { e->err = std::string("sg_latency_probe: ") + hipGetErrorString(r); return SG_ENODEV; }
    *ns_per_load = 10.0 * (double)h[0] / (double)steps;
    return SG_OK;
}

// Shard an event is routed to: owner of its from-endpoint after the join and the optional
// ReverseDirection — exactly what K1 checks.  Host-side, uses the host mirror of the join tables.
int sg_route(sg_handle e, const sg_event* ev, size_t n, uint32_t world, uint32_t* shard_out) {
    if (!e || (!ev && n) || !shard_out || world == 0) return SG_EINVAL;
    std::lock_guard<std::mutex> g(e->mu);
    for (size_t i = 0; i < n; i++) {
        const sg_event& x = ev[i];
        u32 owner;
        auto sp = e->jt.pod_ip.find(x.saddr);
        if (sp == e->jt.pod_ip.end()) { shard_out[i] = sg_fmix32(x.saddr) % world; continue; }   // will be dropped wherever it lands
        owner = owner_hash_ref(SG_MAKE_REF(SG_REF_KNOWN, sp->second));
        if ((x.flags & SG_EV_REVERSE) && !(x.flags & SG_EV_ALIVE)) {    // (K1 never reverses an alive record)
            auto ds = e->jt.svc_ip.find(x.daddr);
            if (ds != e->jt.svc_ip.end()) owner = owner_hash_ref(SG_MAKE_REF(SG_REF_KNOWN, ds->second));
            else {
                auto dp = e->jt.pod_ip.find(x.daddr);
                if (dp != e->jt.pod_ip.end()) owner = owner_hash_ref(SG_MAKE_REF(SG_REF_KNOWN, dp->second));
                else if (x.host_label) owner = owner_hash_ref(SG_MAKE_REF(SG_REF_LABEL, x.host_label - 1));
                else owner = owner_hash_obip(x.daddr);
            }
        }
        shard_out[i] = owner % world;
    }
    return SG_OK;
}

}  // extern "C"
