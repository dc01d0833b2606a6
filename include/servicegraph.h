/*
 * servicegraph.h — C ABI of the MI355X-native ServiceGraph engine.
 *
 * This is the drop-in boundary for the hot path of getanteon/alaz:
 *
 *   L7Event -> processL7 -> process<Proto>Event -> setFromToV2 -> ds.PersistRequest
 *   (aggregator/data.go:310-337, 1364-1383, 1208-1249, 827-870; datastore/backend.go:819-847)
 *
 * The reference resolves every L7 event to an edge (FromUID -> ToUID) on the CPU and ships one
 * 16-field row per request.  The engine behind this header does the same resolution on the GPU,
 * accumulates per-edge integer statistics, builds a CSR adjacency, runs GraphSAGE-mean message
 * passing and emits one scored row per *edge* per window.
 *
 * Everything here is plain C: pointers, sizes, fixed-width integers.  All functions return
 * 0 (SG_OK) or a negative SG_E* code and never throw.  They may be called from arbitrary OS
 * threads (cgo hands calls to whichever thread runs the goroutine); calls on one handle are
 * serialised internally.  The library never keeps a caller pointer after the call returns
 * (cgo rule: C must not retain Go memory; datastore/backend.go:824-839 copies out likewise).
 *
 * There is NO CPU fallback.  sg_create() fails with SG_ENODEV when no gfx950 device is usable.
 */
#ifndef SERVICEGRAPH_H
#define SERVICEGRAPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_ABI_VERSION 6u   /* 6: delta windows — sg_stats.windows_delta / windows_plain / last_window_new_edges.  5: warm windows — sg_stats.windows_warm / windows_cold, SG_CFG_NO_WARM, sg_set_warm; sg_timing_samples,
                               sg_latency_probe;
                               4: sg_stats.ingest_waits, sg_geometry.pass_a_teams, sg_clock_probe;
                               3: sg_config begins with its own size (a binding compiled against an older, shorter sg_config is
                               detected instead of read past its end; members added later are zero for it), sg_geometry_get;
                               2: sg_edge_out carries p50_us / p99_us, sg_config.flags, sg_window_hist, sg_flush_window_view */

/* ---- return codes ---------------------------------------------------------------------- */
#define SG_OK        0
#define SG_EINVAL   (-22)  /* bad argument                                                   */
#define SG_ENOMEM   (-12)  /* device or host allocation failed                               */
#define SG_ENODEV   (-19)  /* no usable HIP device / HIP runtime error (see sg_last_error)   */
#define SG_ENOSPC   (-28)  /* a capacity in sg_config was exceeded (edges, outbound ips, …)  */
#define SG_EAGAIN   (-11)  /* staging ring full: the batch was dropped and counted           */
#define SG_ESTATE   (-71)  /* call made in the wrong window phase                            */

/* ---- L7 protocol numbers: the BPF enum, ebpf/l7_req/l7.go:19-29 ------------------------- */
#define SG_PROTO_UNKNOWN  0
#define SG_PROTO_HTTP     1
#define SG_PROTO_AMQP     2
#define SG_PROTO_POSTGRES 3
#define SG_PROTO_HTTP2    4
#define SG_PROTO_REDIS    5
#define SG_PROTO_KAFKA    6
#define SG_PROTO_MYSQL    7
#define SG_PROTO_MONGO    8

/* ---- sg_event.flags -------------------------------------------------------------------- */
#define SG_EV_TLS      0x01u /* L7Event.Tls                          ebpf/l7_req/l7.go:402  */
#define SG_EV_REVERSE  0x02u /* AMQP DELIVER / Redis PUSHED_EVENT: ReverseDirection() after
                                the join            aggregator/data.go:1110-1112,1151-1153  */
#define SG_EV_CONSUME  0x04u /* Kafka CONSUME record (informational) data.go:1043-1076       */
#define SG_EV_ALIVE    0x08u /* not a request: one open TCP connection saddr -> daddr, as
                                sendOpenConnection() reports it  aggregator/data.go:1628-1679.
                                Joined like a request but without a Host header (ToUID of an
                                unknown daddr is the IP, :1671-1672); creates / keeps the edge
                                and adds 1 to its `alive` count; status, duration and
                                write_time_ns are ignored and no request is counted.           */

/*
 * One L7 request, packed.  32 bytes, little-endian, naturally aligned.
 * It carries exactly the fields of l7_req.L7Event (ebpf/l7_req/l7.go:396-417) that decide the
 * edge identity and the per-edge statistics; the 1 KiB payload stays on the host.
 *
 *  saddr/daddr  numeric a<<24|b<<16|c<<8|d, as in L7Event.Saddr/Daddr (l7.go:413,415), i.e. the
 *               value IntToIPv4() formats (aggregator/data.go:1751-1767).
 *  host_label   0, or a host-interned id (>=1) of the HTTP "Host:" header value that
 *               parseHttpPayload extracts (data.go:508-531).  Only consulted when daddr is neither
 *               a service nor a pod IP: ToUID = Host header (data.go:851-854).
 *  status       L7Event.Status (HTTP status code, or the 1/2/3 protocol status of
 *               ebpf/c/{postgres,redis,mysql}.c), saturated to 16 bits.
 *  protocol     SG_PROTO_*.
 *  duration_ns  L7Event.Duration, Request.Latency (data.go:1220).
 *  write_time_ns L7Event.WriteTimeNs (kernel monotonic); StartTime is derived as
 *               convertKernelTimeToUserspaceTime()/1e6 (data.go:1219,1740-1743).
 */
typedef struct sg_event {
    uint32_t saddr;
    uint32_t daddr;
    uint32_t host_label;
    uint16_t status;
    uint8_t  protocol;
    uint8_t  flags;
    uint64_t duration_ns;
    uint64_t write_time_ns;
} sg_event;

/* ---- node references -------------------------------------------------------------------- *
 * A node of the service map is what the reference calls (Type, UID): FromType/FromUID,
 * ToType/ToUID (datastore/dto.go:177-195).  Across this ABI a node is a tagged 32-bit ref:
 *   KNOWN  pod or service; payload = the id the host passed to sg_upsert_pod/_service
 *          (the host interns UID -> id in arrival order).
 *   LABEL  outbound, named by a Host header; payload = host_label - 1.
 *   OBIP   outbound, named by the raw destination IP (data.go:862-863); payload = index into
 *          the window's ascending outbound-IP list (sg_window_outbound_ips).
 */
#define SG_REF_TYPE(r)    ((uint32_t)(r) >> 30)
#define SG_REF_VALUE(r)   ((uint32_t)(r) & 0x3FFFFFFFu)
#define SG_REF_KNOWN      0u
#define SG_REF_LABEL      1u
#define SG_REF_OBIP       2u
#define SG_MAKE_REF(t, v) (((uint32_t)(t) << 30) | ((uint32_t)(v) & 0x3FFFFFFFu))

/* node kinds stored per KNOWN id */
#define SG_NODE_POD       1u
#define SG_NODE_SERVICE   2u

/*
 * One scored edge of a closed window.  Rows come out sorted by (dense(from), dense(to)) where
 * dense() orders KNOWN ids first, then LABEL, then OBIP — the canonical CSR order.
 *
 *  count/err_count/sum_ns/max_ns/sumsq_us  integer accumulators over the window's events on this
 *        edge: bit-exact.  "error" = HTTP/HTTP2 status >= 500, or status == 2 for
 *        POSTGRES/REDIS/MYSQL (ebpf/c/postgres.c:91, redis.c:10, mysql.c:36).
 *        sumsq_us = sum of (duration_ns / 1000)^2, wrapping u64.
 *  score      GraphSAGE + factorised MLP anomaly score in (0,1)          (fp32, |d| <= 1e-5)
 *  lat_z      (mean_us(edge) - mean_us(src out-events)) / max(std_us(src), 1)  (fp32)
 *  err_ratio  err_count / count                                            (fp32)
 *  p50_us / p99_us  latency percentiles from the edge's log2 histogram, 0 unless the engine was created with
 *        SG_CFG_EDGE_HISTOGRAM.  Bin of a duration d (ns): 0 for d < 2^17 (131 us), k = floor(log2 d) - 16 for
 *        2^17 <= d < 2^31, 15 for d >= 2^31 (2.1 s): one bin per octave.  The q-th percentile is reported as the upper
 *        edge of the first bin whose cumulative count reaches ceil(count * q / 100) — 2^(17+k) ns, max_ns for the open
 *        last bin — capped at max_ns, in microseconds (floor); 0 for an edge without requests.  Integer arithmetic
 *        throughout: bit-exact.  The bins themselves: sg_window_hist().
 */
#define SG_HIST_BINS 16u
typedef struct sg_edge_out {
    uint64_t sum_ns;
    uint64_t max_ns;
    uint64_t sumsq_us;
    uint32_t from_ref;
    uint32_t to_ref;
    uint32_t count;
    uint32_t err_count;
    float    score;
    float    lat_z;
    float    err_ratio;
    uint32_t alive;             /* open connections reported on this edge in the window (SG_EV_ALIVE) */
    uint32_t p50_us;
    uint32_t p99_us;
} sg_edge_out;

typedef struct sg_config {
    uint32_t struct_size;       /* sizeof(sg_config) as the CALLER compiled it.  sg_create accepts any size from the ABI-3
                                   layout (88 bytes) up to its own; members beyond the caller's size read as zero         */
    uint32_t abi_version;       /* SG_ABI_VERSION                                              */
    int32_t  device;            /* HIP device ordinal                                          */
    uint32_t max_known_nodes;   /* capacity of the pod+service id space                        */
    uint32_t max_labels;        /* capacity of the Host-header label space                     */
    uint32_t max_outbound_ips;  /* distinct raw-IP outbound destinations per window            */
    uint32_t max_ips;           /* distinct pod+service IPs in the join tables                 */
    uint64_t max_edges;         /* distinct edges per window                                   */
    uint32_t max_batch;         /* largest n accepted by one sg_ingest()                       */
    uint32_t layers;            /* GraphSAGE layers L, 1..SG_MAX_LAYERS                         */
    uint32_t rank;              /* this shard                                                   */
    uint32_t world;             /* number of shards (1 = unsharded)                             */
    uint32_t k1_variant;        /* 0 = auto: partitioned LDS aggregation when the graph fits it — 8-byte records sorted by
                                       partition in LDS before they are written, from max_edges >= 2^18 or max_window_events
                                       > 2^21 up; the 16-byte-record form of 2 for smaller windows, with SG_CFG_EDGE_HISTOGRAM
                                       and for node spaces beyond 2^24,
                                   1 = global edge table + device-scope atomics (any size),
                                   2 = partitioned aggregation with 16-byte records (the round-2 kernels),
                                   3 = as 0, but the 8-byte form whatever the window's size (where the graph fits it)  */
    uint64_t max_window_events; /* most events one window may carry (sizes the K1 record slabs;
                                   0 = max_batch)                                               */
    uint32_t windows_in_flight; /* 1..8 window slots, each with its own buffers and HIP stream:
                                   sg_window_run() closes the current window on its slot's stream
                                   and moves on to the next slot, so the (latency-bound) close of
                                   window w overlaps the ingest of window w+1.  0 = 1.              */
    uint32_t max_alive;         /* most SG_EV_ALIVE records one window may carry (0 = 65536)        */
    uint32_t flags;             /* SG_CFG_*                                                          */
} sg_config;

#define SG_CFG_EDGE_HISTOGRAM 0x1u /* keep a 16-bin log2 latency histogram per edge (p50_us / p99_us in the rows, the bins through
                                      sg_window_hist).  Costs LDS in both K1 passes (smaller edge cache, half-size partitions) and
                                      64 bytes per edge of extra traffic: off by default.                                          */

#define SG_CFG_NO_WARM 0x2u        /* never carry the edge set from one window to the next: every window is rebuilt from nothing (see
                                      sg_set_warm).  By default an engine on the 8-byte-record path without the histogram and with
                                      max_edges >= 2^18 keeps the union of the edges its windows have touched, in CSR order: a
                                      window whose edges are all among the kept ones skips the degree count, the row scan, the
                                      scatter and the row sort.  The rows of a window are the same either way, bit for bit.         */
#define SG_CFG_WARM    0x4u        /* keep that state below 2^18 edges too (where it does not pay: for tests)                       */

#define SG_MAX_LAYERS 4u
#define SG_F_IN    32u   /* node feature width                                                  */
#define SG_F_HID   64u   /* hidden width of every SAGE layer and of the score head               */
#define SG_F_EDGE   8u   /* edge feature width                                                  */
#define SG_NODE_STAT_SUM_WORDS 12u /* u64 words per node in the SUM-reduced stats block           */
#define SG_NODE_STAT_MAX_WORDS  2u /* u64 words per node in the MAX-reduced stats block           */

typedef struct sg_stats {
    uint64_t events_in;            /* events handed to K1 since create                         */
    uint64_t events_dropped_src;   /* saddr not a pod IP (data.go:829-832)                     */
    uint64_t events_dropped_ring;  /* SG_EAGAIN drops                                          */
    uint64_t events_dropped_cap;   /* edge / outbound-ip capacity overflow                     */
    uint64_t windows;              /* closed windows                                           */
    uint64_t last_window_events;   /* accepted events in the last closed window                */
    uint64_t last_window_edges;
    uint64_t last_window_nodes;
    int64_t  last_window_tmin_ms;  /* min/max Request.StartTime in the window (data.go:1219)    */
    int64_t  last_window_tmax_ms;
    uint64_t h2d_bytes;
    uint64_t events_misrouted;     /* world > 1: events fed to the wrong shard (see sg_route)    */
    uint64_t halo_overflow;        /* halo requests beyond the per-pair capacity (must be 0)     */
    uint64_t alive_in;             /* SG_EV_ALIVE records handed to K1 since create               */
    uint64_t alive_dropped;        /* ... of which beyond max_alive, or with an endpoint that was dropped */
    uint64_t join_word_updates;    /* join-table words changed in place on the device (incremental upserts/deletes) */
    uint64_t join_full_uploads;    /* whole join-table images uploaded (first build, rebuilds)        */
    uint64_t ingest_waits;         /* sg_ingest / sg_ingest_pinned calls that found a window boundary being marked (sg_flush_begin) and
                                      waited for it — bounded by the staging copies in flight; the reference's PersistRequest blocks on a
                                      full channel instead (datastore/backend.go:844), this is the only wait on the aggregator's thread */
    uint64_t windows_warm;         /* of the windows READ so far (sg_flush_* / sg_window_read): closed on the warm path ...             */
    uint64_t windows_cold;         /* ... by the full rebuild (always, for an engine that keeps no state: then both stay 0)            */
    uint64_t windows_delta;        /* of windows_warm: windows that met edges the kept set lacked and merged them in (ABI 6)           */
    uint64_t windows_plain;        /* windows of a state-keeping engine closed WITHOUT touching the kept state (the host's back-off after
                                      repeated fall-backs, raw-outbound-IP streams): counted in neither windows_warm nor windows_cold    */
    uint64_t last_window_new_edges;/* edges the last read window added to the kept set                                                  */
} sg_stats;

typedef struct sg_engine* sg_handle;

/* ---- lifecycle -------------------------------------------------------------------------- */
int  sg_create(const sg_config* cfg, sg_handle* out);
int  sg_destroy(sg_handle h);
const char* sg_last_error(sg_handle h);          /* thread-unsafe diagnostic string             */
uint32_t sg_abi_version(void);
size_t   sg_weights_count(uint32_t layers);      /* number of fp32 values sg_load_weights wants  */

/* ---- join tables: replaces ClusterInfo.PodIPToPodUid / ServiceIPToServiceUid ------------- *
 * (aggregator/cluster.go:13-17) and their maintenance in processPod / processSvc
 * (aggregator/persist.go:55-71, 114-130).  ADD and UPDATE are both "upsert"; DELETE erases the
 * IP.  Pods without an IP are skipped by the caller (persist.go:37-40).                        */
int sg_upsert_pod(sg_handle h, uint32_t ip, uint32_t node_id);
int sg_delete_pod(sg_handle h, uint32_t ip);
int sg_upsert_service(sg_handle h, uint32_t ip, uint32_t node_id);
int sg_delete_service(sg_handle h, uint32_t ip);

/* FirstKernelTime / FirstUserspaceTime (ebpf/l7_req/l7.go:707-710) for StartTime conversion.   */
int sg_set_clock(sg_handle h, uint64_t first_kernel_ns, uint64_t first_user_ns);

/* Number of Host-header labels the host packer has interned so far (labels are cumulative).
 * The engine also tracks the largest label id it has seen; the larger of the two sizes the
 * LABEL id range of the next closed window.                                                    */
int sg_set_label_count(sg_handle h, uint32_t n_labels);

/* fp32 weight blob, layout documented in DESIGN.md §"weights"; copied.                         */
int sg_load_weights(sg_handle h, const float* w, size_t n);

/* ---- ingest: replaces processL7 .. setFromToV2 .. PersistRequest for the edge fields ------ *
 * sg_ingest copies n events from host memory into the engine's pinned staging ring, uploads
 * them (in 4 MiB pieces, each on the link while the caller's thread copies the next) and
 * launches K1 asynchronously.  Non-blocking: a full ring drops the batch, counts it
 * and returns SG_EAGAIN (the reference's PersistRequest would block here, backend.go:844).      */
int sg_ingest(sg_handle h, const sg_event* events, size_t n);

/* The same without the staging copy, for events that already sit in page-locked host memory the caller registered with
 * sg_host_register (a C-allocated buffer a packer writes into; NOT Go-heap memory): the records are read asynchronously (H2D copy,
 * then K1 pass A) and must stay unchanged until sg_flush_end* / sg_flush_window* of their window has RETURNED (sg_flush_begin and
 * sg_window_run only enqueue: after sg_window_run synchronise its stream first).  Non-blocking like sg_ingest (SG_EAGAIN when no device
 * slot is free); SG_EINVAL when the events are not inside registered memory.                                              */
int sg_host_register(sg_handle h, void* p, size_t bytes);
int sg_host_unregister(sg_handle h, void* p);
int sg_ingest_pinned(sg_handle h, const sg_event* events, size_t n);
/* Blocking convenience for loaders that would rather wait than drop (replay tools, benchmarks): n events in max_batch-sized
 * pieces through sg_ingest (pinned = 0) or sg_ingest_pinned (pinned = 1); a full ring is waited for, not counted as a drop.
 * *retries (may be NULL) = how often it had to wait.  The aggregator-facing entry points above stay non-blocking.        */
int sg_ingest_bulk(sg_handle h, const sg_event* events, size_t n, int pinned, uint64_t* retries);

/* Same, for events already resident in device memory (bench, sharded feeder).  `stream` is a
 * hipStream_t (NULL = the engine's own stream); K1 is ordered after prior work of that stream. */
int sg_ingest_device(sg_handle h, const sg_event* d_events, size_t n, void* stream);

/* ---- window close ------------------------------------------------------------------------ *
 * One call for the unsharded case: K2..K5, copy-out, reset.  `out` receives up to `cap` rows,
 * *n the number of edges of the window (may exceed cap; then only cap rows were written).       */
int sg_flush_window(sg_handle h, uint64_t window_end_ms, sg_edge_out* out, size_t cap, size_t* n);

/* The same window close without the copy into caller memory: the rows are transferred into page-locked host memory the
 * engine owns and *rows points at them — valid until the next sg_flush_window / sg_flush_window_view / sg_window_read on
 * this handle, or sg_destroy.  *n = edges of the window (all of them are there).  For callers that only walk the rows once
 * (a Go shim building its payload, GraphDS::FlushWindow): a pageable 64 MB destination costs more than the transfer. */
int sg_flush_window_view(sg_handle h, uint64_t window_end_ms, const sg_edge_out** rows, size_t* n);

/* The window close in two halves, for hosts whose feeders keep running (the aggregator's worker goroutines do): sg_flush_begin
 * marks the window boundary — it waits for the staging copies that began before it (at most one batch copy per feeder; sg_ingest
 * calls that arrive meanwhile wait that long too, then belong to the NEXT window), enqueues K1 pass B .. K5 and returns.
 * sg_flush_end / sg_flush_end_view wait for the pipeline and fetch the rows without holding the engine lock, on a stream of
 * their own: the rows leave over PCIe while the next window's events arrive (full duplex), and no feeder stands still for the
 * ~1.7 ms a C3 window's kernels + copy-out take.  One begin may be open at a time (SG_ESTATE otherwise); any thread may call
 * the end.  sg_flush_window / sg_flush_window_view are exactly begin + end.                                                   */
int sg_flush_begin(sg_handle h, uint64_t window_end_ms);
int sg_flush_end(sg_handle h, sg_edge_out* out, size_t cap, size_t* n);
int sg_flush_end_view(sg_handle h, const sg_edge_out** rows, size_t* n);

/* Enqueue-only form of the same pipeline (K2..K5 + reset, no copy-out, no host sync) for
 * callers that keep results on the device (sg_window_rows_buffer) or time the pipeline.         */
int sg_window_run(sg_handle h, void* stream);
int sg_window_rows_buffer(sg_handle h, void** d_rows);   /* device sg_edge_out[max_edges] of the window
                                                             sg_window_run closed last (valid until its slot is reused) */

/* The same pipeline in stages, so that a sharded driver can run its exchanges in between.
 * Order: close -> [allreduce node stats] -> features -> for l in 0..L-1 { layer(l) ->
 * [halo exchange of layer l+1 rows] } -> score -> read -> reset.
 * All stage calls enqueue on `stream` (NULL = engine stream) and do not synchronise.            */
int sg_window_close(sg_handle h, void* stream);      /* K2: canonical ids, CSR; K3a: partial node stats */
/* Sharded close: OBIP numbering must agree on every shard, so the driver gathers every shard's
 * raw outbound IPs (sg_window_obip_list fills a caller-owned device list + device count), concatenates
 * them into d_union_ips (device, capacity >= next_pow2(world * max_outbound_ips), duplicates
 * allowed) and passes the total count in *d_union_n (device).                                   */
int sg_window_obip_list(sg_handle h, uint32_t* d_list, uint32_t cap, uint32_t* d_n, void* stream);
int sg_window_close_sharded(sg_handle h, const uint32_t* d_union_ips, const uint32_t* d_union_n, void* stream);
/* Sharded close with no host round trip: d_gathered is the all-gather of every shard's
 * [count, ip, ip, ...] u32 buffer (sg_window_obip_list with d_list = buf + 1, d_n = buf), `stride` u32
 * per shard.                                                                                     */
int sg_window_close_gathered(sg_handle h, const uint32_t* d_gathered, uint32_t stride, uint32_t world, void* stream);
int sg_window_features(sg_handle h, void* stream);   /* K3b: node + edge features from reduced stats    */
int sg_window_layer(sg_handle h, uint32_t l, void* stream);  /* K4: rows with out-edges owned here + all rows without out-edges */
int sg_window_score(sg_handle h, void* stream);      /* K5 */
int sg_window_score_reset(sg_handle h, void* stream);/* K5 with the window reset folded in (one launch less): for drivers
                                                        that read the rows through sg_window_rows_buffer(), not
                                                        sg_window_read(); the window is open again afterwards      */
int sg_window_read(sg_handle h, sg_edge_out* out, size_t cap, size_t* n); /* device-syncs, copies out; also valid after
                                                        sg_window_score_reset / sg_window_run_sharded until the next ingest */
int sg_window_reset(sg_handle h, void* stream);      /* clears the window state                  */

/* Device buffers a sharded driver reduces / exchanges (all device pointers, engine-owned):
 *  stats_sum: [n_nodes_cap][SG_NODE_STAT_SUM_WORDS] u64, SUM-reduce across shards
 *  stats_max: [n_nodes_cap][SG_NODE_STAT_MAX_WORDS] u64, MAX-reduce across shards
 *  counters : [8] u64 — {n_known, n_labels, n_obip, n_edges, n_events, dropped_src, dropped_cap, 0}
 *  feat(l)  : [n_nodes_cap][SG_F_HID] fp32 rows of layer l output (l = 1..L); l = 0 gives the
 *             [n_nodes_cap][SG_F_IN] input features.                                           */
int sg_window_buffers(sg_handle h, void** stats_sum, void** stats_max, void** counters,
                      size_t* n_nodes_cap);
int sg_window_feat_buffer(sg_handle h, uint32_t l, void** rows, size_t* row_floats);

/* Caller-owned device memory for the buffers a sharded driver reduces / exchanges in place
 * (stats_sum [ncap][SG_NODE_STAT_SUM_WORDS] u64, stats_max [ncap][2] u64, feat_rows[l] = layer l+1 rows [ncap][64] f32).
 * NULL entries keep the engine's own buffer.                                                    */
int sg_bind_buffers(sg_handle h, void* stats_sum, void* stats_max, void* const* feat_rows, uint32_t n_feat);

/* Halo support (K6).  Fill `ids` (device, u32[cap]) with the dense node indices this shard needs
 * from other shards — destinations of local edges that are not owned here and have out-edges —
 * grouped by owner shard (ascending id inside a group); counts[k] (device, u32[world]) = number
 * of ids owned by shard k.  pack/unpack move rows of layer l between the feature buffer and a
 * contiguous exchange buffer.  Call after the node statistics have been reduced.                */
int sg_halo_build(sg_handle h, uint32_t* d_ids, uint32_t cap, uint32_t* d_counts, void* stream);
/* Padded variants for a fixed-size all-to-all (no host synchronisation): lists are [world][capp + 1]
 * u32, element 0 = count.  Requests beyond capp are dropped and counted in sg_stats.halo_overflow.   */
int sg_halo_build_padded(sg_handle h, uint32_t* d_req, uint32_t capp, void* stream);
int sg_halo_pack_padded(sg_handle h, uint32_t l, const uint32_t* d_serve, uint32_t capp, float* d_rows, void* stream);
int sg_halo_unpack_padded(sg_handle h, uint32_t l, const uint32_t* d_req, uint32_t capp, const float* d_rows, void* stream);
int sg_halo_pack(sg_handle h, uint32_t l, const uint32_t* d_ids, uint32_t n, float* d_rows, void* stream);
int sg_halo_unpack(sg_handle h, uint32_t l, const uint32_t* d_ids, uint32_t n, const float* d_rows, void* stream);

/* ---- the sharded window in ONE call ------------------------------------------------------------------------------------ *
 * sg_window_run_sharded = sg_window_obip_list .. sg_window_score_reset above with every exchange in between — all-gather of
 * the raw outbound IPs, SUM / MAX all-reduce of the integer node statistics, all-to-all of the halo request lists, per layer
 * an all-to-all of exactly the requested rows — issued by the library itself on RCCL (grouped ncclSend / ncclRecv over the
 * xGMI full mesh) and enqueued on `stream`: one C call per window, nothing waits for the device, the exchange buffers belong
 * to the engine.  The communicator: rank 0 calls sg_comm_unique_id, the caller broadcasts the 128 bytes (any transport),
 * every rank calls sg_comm_create with its sg_config.rank / world.  RCCL is dlopen'ed (SG_ENODEV if it cannot be).      */
typedef struct sg_comm sg_comm;
int sg_comm_probe(void);                         /* SG_OK when librccl can be loaded with every symbol the library uses: lets all ranks agree
                                                    on a fallback BEFORE any of them enters ncclCommInitRank (which would wait for the others) */
int sg_comm_unique_id(void* id128, size_t bytes);
int sg_comm_create(const void* id128, size_t bytes, int rank, int world, int device, sg_comm** out);
int sg_comm_destroy(sg_comm* c);
int sg_window_run_sharded(sg_handle h, sg_comm* comm, void* stream);
/* Rows this shard asked each owner rank for in its last sharded window (counts[r], r < world).  Diagnostic; device-syncs. */
int sg_window_halo_counts(sg_handle h, uint32_t* counts, size_t world);

/* Ascending raw IPs of the last read window's OBIP nodes.                                       */
int sg_window_outbound_ips(sg_handle h, uint32_t* ips, size_t cap, size_t* n);
/* The latency histograms of the last read window (SG_CFG_EDGE_HISTOGRAM): bins[i * SG_HIST_BINS + b] = requests of row i in
 * bin b, rows in the order sg_flush_window / sg_window_read returned them.  *n = rows available.  SG_ESTATE without the flag. */
int sg_window_hist(sg_handle h, uint32_t* bins, size_t cap_rows, size_t* n);

int sg_stats_get(sg_handle h, sg_stats* out);

/* What sg_create chose for K1 (diagnostic: bench.py and the tuning scripts name the kernels they time with it). */
typedef struct sg_geometry {
    uint32_t k1_variant;        /* 0 partitioned aggregation, 1 global table + atomics                                      */
    uint32_t k1_narrow;         /* variant 0: 1 = 8-byte records (k1a_tile_partition / k1b_stream_merge), 0 = 16-byte records */
    uint32_t partitions;        /* np                                                                                        */
    uint32_t table_slots;       /* pass B: LDS table slots per partition                                                     */
    uint32_t pass_a_workgroups; /* pieces per partition                                                                      */
    uint32_t cache_slots;       /* pass A: LDS edge-cache slots (follows the join tables' size)                              */
    uint32_t join_l2_in_lds;    /* pass A: level 2 of the join staged in LDS (1) or read from global memory (0)              */
    uint32_t tile_records;      /* narrow: records sorted per tile                                                           */
    uint32_t endpoint_bits;     /* narrow: nb; remainder bits = 2 nb - log2(partitions)                                      */
    uint32_t piece_bytes;       /* record slab bytes per (partition, workgroup) piece                                        */
    uint32_t pass_b_split;      /* narrow: pass-B workgroups (sub-tables) per partition                                      */
    uint32_t pass_a_teams;      /* narrow: k1a_team_partition with 2 teams of eight waves per workgroup or 1 team of sixteen; 0 = k1a_tile_partition */
    uint32_t warm_windows;      /* 1 = the engine carries the edge set and its CSR order from window to window (SG_CFG_NO_WARM)   */
} sg_geometry;
int sg_geometry_get(sg_handle h, sg_geometry* out);

/* Per-kernel timing, measured on the launch stream.  Groups: 1 = K1 pass A (k1a_partition / k1_resolve_aggregate, one record
 * per batch), 7 = K1 pass B (k1b_merge) — both by the dispatch's own begin/end stamps; 2 = K2 csr_build (two records per window:
 * window bookkeeping, then row pointers + scatter + row sort), 8 = K3 in-statistics, 3 = K3 node + edge features, 4 = K4 (one
 * record per SAGE layer), 5 = K5, 6 = K6 halo kernels, 9 = the collectives of sg_window_run_sharded (one record per RCCL call) — by hipEvent
 * pairs around the launches.  sg_timing_get returns the
 * average duration in microseconds per record since sg_timing_reset(), and the number of records.  An engine that keeps warm-window
 * state launches pass B twice per window (the warm attempt and the cold merge; one of them returns at once): group 7 then has two
 * records per window and a window's pass B is their sum.                                                                    */
int sg_timing_enable(sg_handle h, int on);   /* 0 = off, 1 = every group, else bitmask: bit k = group Kk */
int sg_timing_reset(sg_handle h);
int sg_timing_stride(sg_handle h, uint32_t n);   /* groups 1 and 7 (dispatch stamps, a few us per launch): on every n-th window only; 1 = every window */
int sg_timing_get(sg_handle h, int kernel, double* avg_us, uint64_t* launches);
/* Warm windows on / off at run time (on = 0: no window tries the warm path from now on, each is rebuilt and re-captured; on = 1: back
 * to the default).  The rows never depend on it; bench.py uses it to time the cold path beside the steady state.                     */
int sg_set_warm(sg_handle h, int on);
/* Every record of a group since sg_timing_reset, in launch order: min(*n, cap) durations in microseconds go to us[], *n = how many
 * there are.  Group 10 (only when its bit is set explicitly or with on = 1) = one record per window of sg_window_run /
 * sg_window_run_sharded, from in front of the window's first pass-A launch to behind its score kernel.                      */
int sg_timing_samples(sg_handle h, int kernel, double* us, size_t cap, size_t* n);
/* Memory latency of this box: one lane follows `steps` dependent loads (one 128-byte line each, an odd-multiplier walk over all
 * lines) through `bytes` of device memory it allocates for the call; *ns_per_load by the 100 MHz reference clock.  warm & 1
 * walks every line once before the clock starts (a 2 MiB buffer then measures the L2, a buffer far beyond the 256 MiB Infinity
 * Cache without it measures HBM); warm & 2: 65 536 lanes follow a chain each at the same time and one of them is timed — the
 * latency of a random line under load.  Diagnostic for bench.py ("which kind of box did this line come from"); device-syncs.  */
int sg_latency_probe(sg_handle h, uint64_t bytes, uint32_t steps, int warm, double* ns_per_load);
/* The shader clock the chip sustains, in MHz (shader cycles per 100 MHz reference tick x 100): *spin_mhz from an all-CU integer
 * spin of about spin_us microseconds launched by this call, *k1a_mhz averaged over the K1 pass-A launches since the previous
 * call (0 if none; narrow-record kernels only).  Diagnostic for bench.py: the boxes of a pool differ in the clock they hold
 * under load.  Device-syncs.                                                                                               */
int sg_clock_probe(sg_handle h, uint32_t spin_us, double* spin_mhz, double* k1a_mhz);
/* Tuning aid: with SG_ABLATE & 0x100 in the environment at sg_create, the K1 kernels record 100 MHz
 * wall-clock stamps at their phase boundaries, [kernel 0..3][4096 workgroups][8 stamps] u64; this
 * copies the first n words out.  All zero otherwise.                                                */
int sg_debug_stamps(sg_handle h, uint64_t* out, size_t n);

/* Owner shard of a node / routing shard of an event: murmur3 fmix32(ip) % world.
 * The feeder routes an event by the IP of its from-endpoint: daddr if SG_EV_REVERSE (and not SG_EV_ALIVE) else saddr. */
uint32_t sg_hash32(uint32_t x);
/* Shard each event must be fed to when world > 1: owner of its from-endpoint after the join and
 * the optional direction reversal — the same rule K1 enforces (misrouted events are dropped and
 * counted).  Host-only: uses the host mirror of the join tables, launches nothing.              */
int sg_route(sg_handle h, const sg_event* events, size_t n, uint32_t world, uint32_t* shard_out);

#ifdef __cplusplus
}
#endif
#endif /* SERVICEGRAPH_H */
