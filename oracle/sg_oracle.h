/*
 * sg_oracle.h — CPU oracle for the ServiceGraph hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (alaz_amd/, include/servicegraph.h) never links, imports or calls it.
 *
 * PARITY STATUS
 *   edge identity / latency / status / ReqInfo layout:
 *       restated from the reference source (file:line cited at each function in sg_oracle.c).
 *       The reference cannot be built here (no Go toolchain, no vendored modules) and its own
 *       tests hold no golden vector for this path, so the restatement is
 *       pinned by source reading plus the adjacent known-answer tests that do exist
 *       (aggregator/pg_test.go Parse/Bind strings; the simulator constants of
 *       main_benchmark_test.go:562,585-598) — see tests/golden/.
 *   scores (GraphSAGE + MLP):  PARITY UNPINNED — the reference has no scoring code at all;
 *       this file *defines* the model and tests/ cross-check it against an
 *       independent numpy implementation (oracle/score_np.py).
 */
#ifndef SG_ORACLE_H
#define SG_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/servicegraph.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OR_L7_WIRE_SIZE 1096   /* sizeof(struct l7_event), ebpf/c/l7.c:19-47; l7.go:345-369 */
#define OR_UID_MAX      160

typedef struct oracle oracle_t;

/* One row as BackendDS.PersistRequest lays it out (datastore/backend.go:824-839; ReqInfo is
 * [16]interface{}, datastore/payload.go:109-125).  Kafka rows (backend.go:854-869) reuse the
 * struct: protocol="KAFKA", method = Type ("PUBLISH"/"CONSUME"), path = Topic. */
typedef struct or_reqinfo {
    int64_t  start_time;        /* [0]  ms                           */
    uint64_t latency;           /* [1]  ns                           */
    char     from_ip[16];       /* [2]                               */
    char     from_type[10];     /* [3]  "pod"|"service"|"outbound"   */
    char     from_uid[OR_UID_MAX]; /* [4]                            */
    uint16_t from_port;         /* [5]                               */
    char     to_ip[16];         /* [6]                               */
    char     to_type[10];       /* [7]                               */
    char     to_uid[OR_UID_MAX];/* [8]                               */
    uint16_t to_port;           /* [9]                               */
    char     protocol[10];      /* [10]                              */
    uint32_t status_code;       /* [11]                              */
    char     fail_reason[4];    /* [12] always ""                    */
    char     method[24];        /* [13]                              */
    char     path[1100];        /* [14]                              */
    uint8_t  tls;               /* [15]                              */
    uint8_t  is_kafka;
} or_reqinfo;

typedef struct or_edge {
    sg_edge_out row;            /* same layout the engine emits      */
    char from_type[10], to_type[10];
    char from_uid[OR_UID_MAX], to_uid[OR_UID_MAX];
} or_edge;

oracle_t* or_create(void);
void      or_destroy(oracle_t* o);

/* l7_req.FirstKernelTime / FirstUserspaceTime (l7.go:333-334, 707-710). */
void or_set_clock(oracle_t* o, uint64_t first_kernel_ns, uint64_t first_user_ns);
/* keep at most `limit` full ReqInfo rows (0 = none); aggregation is unaffected. */
void or_set_log_limit(oracle_t* o, size_t limit);

/* aggregator/persist.go:25-72 (processPod) and :81-131 (processSvc). event_type "ADD"|"UPDATE"|"DELETE".
 * Empty pod_ip => skipped (persist.go:37-40). Returns the interned node id, or -1 if skipped. */
int or_process_pod(oracle_t* o, const char* event_type, const char* uid, const char* pod_ip);
int or_process_svc(oracle_t* o, const char* event_type, const char* uid, const char* cluster_ip);

/* n wire records of OR_L7_WIRE_SIZE bytes each, through processL7 (data.go:1364-1383).
 * kafka_msgs: optional per-record count of decoded Kafka messages (NULL => 1 for every KAFKA
 * record); the Sarama-derived decoder itself is out of scope.
 * Returns the number of rows that reached PersistRequest/PersistKafkaEvent. */
size_t or_process_l7_wire(oracle_t* o, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs);

/* The same join + persist for already-packed events (the payload-dependent work — Host header
 * interning, SQL/Mongo drop decisions, Kafka fan-out — was done by the packer).
 * labels[i] is the Host header string of host_label i+1. */
size_t or_process_packed(oracle_t* o, const sg_event* ev, size_t n,
                         const char* const* labels, size_t n_labels);

size_t or_reqinfo_count(const oracle_t* o);          /* rows persisted (also beyond the log limit) */
size_t or_reqinfo_logged(const oracle_t* o);
const or_reqinfo* or_reqinfo_at(const oracle_t* o, size_t i);
uint64_t or_dropped_src(const oracle_t* o);           /* setFromToV2 returned error (data.go:829-832) */
uint64_t or_dropped_parse(const oracle_t* o);         /* payload parser returned error               */

/* host-side label interning state, as the packer would hold it */
size_t      or_label_count(const oracle_t* o);
const char* or_label_at(const oracle_t* o, size_t i);
size_t      or_known_count(const oracle_t* o);

/* Close the window: build edges (sorted canonically), CSR, node stats, features, L SAGE layers,
 * scores.  weights: blob of sg_weights_count(layers) floats.
 * Returns number of edges.  Resets the per-window accumulators afterwards (tables persist). */
size_t or_window_close(oracle_t* o, const float* weights, uint32_t layers);
size_t or_edge_count(const oracle_t* o);
const or_edge* or_edge_at(const oracle_t* o, size_t i);
size_t or_node_count(const oracle_t* o);
/* debug/inspection of the last closed window */
const float*    or_node_features(const oracle_t* o);           /* [N][SG_F_IN]           */
const float*    or_layer_output(const oracle_t* o, uint32_t l);/* [N][SG_F_HID], l=1..L  */
const uint64_t* or_node_stats_sum(const oracle_t* o);          /* [N][10]                */
const uint64_t* or_node_stats_max(const oracle_t* o);          /* [N][2]                 */
const uint32_t* or_outbound_ips(const oracle_t* o, size_t* n);
int64_t or_window_tmin(const oracle_t* o);
int64_t or_window_tmax(const oracle_t* o);
uint64_t or_window_events(const oracle_t* o);

/* stand-alone pieces exposed for the known-answer tests */
#define OR_HTTP_TOK_CAP  64
#define OR_HTTP_PATH_CAP 1100
/* method[OR_HTTP_TOK_CAP], path[OR_HTTP_PATH_CAP], version[OR_HTTP_TOK_CAP], host[OR_UID_MAX] */
void   or_parse_http_payload(const char* req, size_t len, char* method, char* path, char* version,
                             char* host);                             /* data.go:508-531  */
int    or_parse_postgres(oracle_t* o, uint32_t pid, uint64_t fd, const char* method,
                         const uint8_t* payload, size_t size, char* out, size_t cap); /* :1474-1556 */
void   or_int_to_ipv4(uint32_t ip, char out[16]);                      /* data.go:1751-1767 */
size_t or_weights_count(uint32_t layers);
uint32_t or_hash32(uint32_t x);

#ifdef __cplusplus
}
#endif
#endif
