/*
 * sg_oracle.c — CPU oracle for the ServiceGraph hot path.  TEST INFRASTRUCTURE ONLY
 * (see sg_oracle.h for who may load it and for the parity-pinning status).
 *
 * Part 1 restates, function by function, what getanteon/alaz does on the CPU for one L7 event
 * (file:line of the reference given at each function).  Part 2 is the definition of the edge
 * aggregation + GraphSAGE scoring model, which the reference does not have.
 *
 * Single-threaded, deterministic, scalar C99.  Built by oracle/Makefile with -ffp-contract=off:
 * every floating-point operation below is exactly the one written (fmaf where fmaf is written).
 */
#define _GNU_SOURCE
#include "sg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small containers                                                                           */
/* ------------------------------------------------------------------------------------------ */

static uint64_t fnv1a(const char* s, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (uint8_t)s[i]; h *= 1099511628211ull; }
    return h;
}

/* string -> (string value, u32 value).  Open addressing with tombstones; Go map semantics. */
typedef struct { char* key; char* sval; uint32_t uval; int state; /*0 empty 1 used 2 dead*/ } sm_ent;
typedef struct { sm_ent* e; size_t cap, used, filled; } strmap;

static void sm_init(strmap* m) { m->cap = 64; m->used = m->filled = 0; m->e = calloc(m->cap, sizeof(sm_ent)); }
static void sm_free(strmap* m) {
    for (size_t i = 0; i < m->cap; i++) if (m->e[i].state == 1) { free(m->e[i].key); free(m->e[i].sval); }
    free(m->e); m->e = NULL;
}
static sm_ent* sm_find(const strmap* m, const char* k) {
    size_t n = strlen(k), i = fnv1a(k, n) & (m->cap - 1);
    for (;;) {
        sm_ent* e = &m->e[i];
        if (e->state == 0) return NULL;
        if (e->state == 1 && strcmp(e->key, k) == 0) return e;
        i = (i + 1) & (m->cap - 1);
    }
}
static void sm_put_raw(strmap* m, char* k, char* sv, uint32_t uv) {
    size_t i = fnv1a(k, strlen(k)) & (m->cap - 1);
    while (m->e[i].state == 1) i = (i + 1) & (m->cap - 1);
    if (m->e[i].state == 0) m->filled++;
    m->e[i].key = k; m->e[i].sval = sv; m->e[i].uval = uv; m->e[i].state = 1; m->used++;
}
static void sm_grow(strmap* m) {
    strmap n; n.cap = m->cap * 2; n.used = n.filled = 0; n.e = calloc(n.cap, sizeof(sm_ent));
    for (size_t i = 0; i < m->cap; i++) if (m->e[i].state == 1) sm_put_raw(&n, m->e[i].key, m->e[i].sval, m->e[i].uval);
    free(m->e); *m = n;
}
static sm_ent* sm_put(strmap* m, const char* k, const char* sv, uint32_t uv) {
    sm_ent* e = sm_find(m, k);
    if (e) { if (sv) { free(e->sval); e->sval = strdup(sv); } e->uval = uv; return e; }
    if ((m->filled + 1) * 2 > m->cap) sm_grow(m);
    sm_put_raw(m, strdup(k), sv ? strdup(sv) : NULL, uv);
    return sm_find(m, k);
}
static void sm_del(strmap* m, const char* k) {
    sm_ent* e = sm_find(m, k);
    if (!e) return;
    free(e->key); free(e->sval); e->key = e->sval = NULL; e->state = 2; m->used--;
}

/* delete(m, key) for every key with the given string prefix (Go: range + strings.HasPrefix + delete) */
static size_t sm_del_prefix(strmap* m, const char* prefix) {
    const size_t pl = strlen(prefix); size_t n = 0;
    for (size_t i = 0; i < m->cap; i++) {
        sm_ent* e = &m->e[i];
        if (e->state == 1 && strncmp(e->key, prefix, pl) == 0) { free(e->key); free(e->sval); e->key = e->sval = NULL; e->state = 2; m->used--; n++; }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* oracle state                                                                               */
/* ------------------------------------------------------------------------------------------ */

typedef struct {             /* one (FromType,FromUID,ToType,ToUID) edge of the open window */
    uint32_t from_ref, to_ref;   /* KNOWN/LABEL refs final; OBIP refs hold the raw IP index below */
    uint32_t from_obip, to_obip; /* raw IP when the endpoint is an OBIP node */
    uint32_t count, err;
    uint64_t sum_ns, max_ns, sumsq_us;
    uint32_t alive;              /* alive connections reported on this edge (f-2) */
    uint32_t hist[SG_HIST_BINS]; /* f-3: log2 latency histogram, bins as defined in include/servicegraph.h */
} w_edge;

struct oracle {
    /* ClusterInfo (aggregator/cluster.go:13-17): IP string -> UID string */
    strmap pod_ip_to_uid, svc_ip_to_uid;
    /* host-side interning, as the GraphDS shim keeps it: UID -> node id */
    strmap uid_to_id;  char** id_to_uid; uint8_t* id_kind; size_t n_known, cap_known;
    strmap label_to_id; char** labels; size_t n_labels, cap_labels;
    /* prepared statements (data.go: pgStmts / mySqlStmts) */
    strmap pg_stmts, mysql_stmts;
    uint64_t first_kernel, first_user;

    /* ReqInfo log */
    or_reqinfo* log; size_t log_n, log_cap, log_limit; size_t persisted;
    uint64_t dropped_src, dropped_parse;

    /* f-2: clusterInfo.SocketMaps (pid -> fd -> SocketLine), flattened to "pid:fd" -> index */
    strmap sock_index; or_sockline** socklines; size_t n_socklines, cap_socklines;   /* a cleared process leaves NULL slots */
    char* proc_root; uint64_t now_user_ns;            /* set: a new line is seeded from <proc_root>/<pid>/... (NewSocketLine fetch = true) */
    or_alive* alive_log; size_t alive_n, alive_cap; size_t alive_persisted;
    or_h2* h2;                /* f-4: HTTP/2 request assembly (http2.c) */
    int kafka_decode;         /* f-4: decode Kafka payloads (kafka.c) instead of taking the message count as a side input */

    /* open window */
    strmap edge_index;       /* "ft\x1fuid\x1ftt\x1fuid" -> index into wedges */
    w_edge* wedges; size_t n_wedges, cap_wedges;
    int64_t tmin, tmax; uint64_t wevents;

    /* last closed window */
    or_edge* edges; size_t n_edges;
    uint32_t* edge_hist;         /* [n_edges][SG_HIST_BINS], row order */
    size_t n_nodes;
    float* x0; float* h[SG_MAX_LAYERS + 1];
    uint64_t* st_sum; uint64_t* st_max;
    uint32_t* obips; size_t n_obips;
    int64_t ctmin, ctmax; uint64_t cevents;
};

oracle_t* or_create(void) {
    oracle_t* o = calloc(1, sizeof(*o));
    sm_init(&o->pod_ip_to_uid); sm_init(&o->svc_ip_to_uid); sm_init(&o->uid_to_id);
    sm_init(&o->label_to_id); sm_init(&o->pg_stmts); sm_init(&o->mysql_stmts); sm_init(&o->edge_index); sm_init(&o->sock_index);
    o->tmin = INT64_MAX; o->tmax = INT64_MIN;
    o->h2 = or_h2_create();
    return o;
}
or_h2* or_h2_of(oracle_t* o) { return o->h2; }
void or_set_kafka_decode(oracle_t* o, int on) { o->kafka_decode = on; }

static void free_closed(oracle_t* o) {
    free(o->edges); o->edges = NULL; o->n_edges = 0;
    free(o->x0); o->x0 = NULL;
    for (unsigned l = 0; l <= SG_MAX_LAYERS; l++) { free(o->h[l]); o->h[l] = NULL; }
    free(o->st_sum); free(o->st_max); o->st_sum = o->st_max = NULL;
    free(o->obips); o->obips = NULL; o->n_obips = 0;
}

void or_destroy(oracle_t* o) {
    if (!o) return;
    sm_free(&o->pod_ip_to_uid); sm_free(&o->svc_ip_to_uid); sm_free(&o->uid_to_id);
    sm_free(&o->label_to_id); sm_free(&o->pg_stmts); sm_free(&o->mysql_stmts); sm_free(&o->edge_index);
    for (size_t i = 0; i < o->n_known; i++) free(o->id_to_uid[i]);
    free(o->id_to_uid); free(o->id_kind);
    for (size_t i = 0; i < o->n_labels; i++) free(o->labels[i]);
    free(o->labels); free(o->log); free(o->wedges);
    sm_free(&o->sock_index);
    for (size_t i = 0; i < o->n_socklines; i++) or_sl_destroy(o->socklines[i]);
    free(o->socklines); free(o->alive_log); free(o->proc_root);
    or_h2_destroy(o->h2);
    free_closed(o);
    free(o);
}

void or_set_clock(oracle_t* o, uint64_t fk, uint64_t fu) { o->first_kernel = fk; o->first_user = fu; }
void or_set_log_limit(oracle_t* o, size_t limit) { o->log_limit = limit; }

/* ------------------------------------------------------------------------------------------ */
/* Part 1 — restatement of the reference                                                      */
/* ------------------------------------------------------------------------------------------ */

/* IntToIPv4(u32).String()  — aggregator/data.go:1751-1758, used by extractAddressPair :1760-1767.
 * binary.BigEndian.PutUint32 => most significant byte first. */
void or_int_to_ipv4(uint32_t ip, char out[16]) {
    snprintf(out, 16, "%u.%u.%u.%u", ip >> 24, (ip >> 16) & 255, (ip >> 8) & 255, ip & 255);
}

/* convertKernelTimeToUserspaceTime — data.go:1740-1743; StartTime = int64(that / 1e6) :1219.
 * u64 wrap-around arithmetic, integer division. */
static int64_t start_time_ms(const oracle_t* o, uint64_t write_ns) {
    uint64_t t = o->first_user - (o->first_kernel - write_ns);
    return (int64_t)(t / 1000000ull);
}

/* host shim interning: UID -> dense id in arrival order (GraphDS.ids). */
static uint32_t intern_uid(oracle_t* o, const char* uid, uint8_t kind) {
    sm_ent* e = sm_find(&o->uid_to_id, uid);
    if (e) { o->id_kind[e->uval] = kind; return e->uval; }
    if (o->n_known == o->cap_known) {
        o->cap_known = o->cap_known ? o->cap_known * 2 : 256;
        o->id_to_uid = realloc(o->id_to_uid, o->cap_known * sizeof(char*));
        o->id_kind = realloc(o->id_kind, o->cap_known);
    }
    uint32_t id = (uint32_t)o->n_known++;
    o->id_to_uid[id] = strdup(uid); o->id_kind[id] = kind;
    sm_put(&o->uid_to_id, uid, NULL, id);
    return id;
}

static uint32_t intern_label(oracle_t* o, const char* host) {
    sm_ent* e = sm_find(&o->label_to_id, host);
    if (e) return e->uval;
    if (o->n_labels == o->cap_labels) {
        o->cap_labels = o->cap_labels ? o->cap_labels * 2 : 64;
        o->labels = realloc(o->labels, o->cap_labels * sizeof(char*));
    }
    uint32_t id = (uint32_t)o->n_labels++;
    o->labels[id] = strdup(host);
    sm_put(&o->label_to_id, host, NULL, id);
    return id;
}

/* processPod — aggregator/persist.go:25-72.  PodIP=="" returns before touching the table (:37-40).
 * ADD and UPDATE both assign (:55-64); DELETE deletes the IP key (:65-69). */
int or_process_pod(oracle_t* o, const char* event_type, const char* uid, const char* pod_ip) {
    if (pod_ip == NULL || pod_ip[0] == 0) return -1;
    uint32_t id = intern_uid(o, uid, SG_NODE_POD);
    if (strcmp(event_type, "ADD") == 0 || strcmp(event_type, "UPDATE") == 0) sm_put(&o->pod_ip_to_uid, pod_ip, uid, 0);
    else if (strcmp(event_type, "DELETE") == 0) sm_del(&o->pod_ip_to_uid, pod_ip);
    return (int)id;
}

/* processSvc — aggregator/persist.go:81-131, keyed on Spec.ClusterIP only (:117,:122,:127).
 * (An empty ClusterIP is stored under the key "" by the reference; harmless, kept.) */
int or_process_svc(oracle_t* o, const char* event_type, const char* uid, const char* cluster_ip) {
    uint32_t id = intern_uid(o, uid, SG_NODE_SERVICE);
    if (strcmp(event_type, "ADD") == 0 || strcmp(event_type, "UPDATE") == 0) sm_put(&o->svc_ip_to_uid, cluster_ip, uid, 0);
    else if (strcmp(event_type, "DELETE") == 0) sm_del(&o->svc_ip_to_uid, cluster_ip);
    return (int)id;
}

/* parseHttpPayload — aggregator/data.go:508-531.
 *   lines = Split(request, "\n"); parts = Split(lines[0], " "); if len(parts) >= 3 {method,path,version}
 *   first later line with prefix "Host:": hostParts = Split(line, " "); if len >= 2 { host =
 *   TrimSuffix(hostParts[1], "\r"); break }  (no break when the line has no space) */
void or_parse_http_payload(const char* req, size_t len, char* method, char* path, char* version, char* host) {
    const size_t caps[3] = { OR_HTTP_TOK_CAP, OR_HTTP_PATH_CAP, OR_HTTP_TOK_CAP };
    const size_t cap = OR_UID_MAX;
    method[0] = path[0] = version[0] = host[0] = 0;
    size_t l0 = 0; while (l0 < len && req[l0] != '\n') l0++;
    /* Split(lines[0], " ") */
    size_t starts[4], ends[4]; int np = 0; size_t s = 0;
    for (size_t i = 0; i <= l0; i++) {
        if (i == l0 || req[i] == ' ') { if (np < 3) { starts[np] = s; ends[np] = i; } np++; s = i + 1; }
    }
    if (np >= 3) {
        char* dst[3] = { method, path, version };
        for (int p = 0; p < 3; p++) {
            size_t n = ends[p] - starts[p]; if (n >= caps[p]) n = caps[p] - 1;
            /* parts[2] is the whole third field only if there are exactly 3 parts; with more parts it
             * is still the third token — Split gives tokens, we mirror that */
            memcpy(dst[p], req + starts[p], n); dst[p][n] = 0;
        }
    }
    size_t pos = l0 + 1;
    while (pos <= len && l0 < len) {
        size_t e = pos; while (e < len && req[e] != '\n') e++;
        size_t n = e - pos;
        if (n >= 5 && memcmp(req + pos, "Host:", 5) == 0) {
            /* Split(line, " "): need at least one space */
            const char* sp = memchr(req + pos, ' ', n);
            if (sp) {
                const char* b = sp + 1; const char* lim = req + e;
                const char* q = b; while (q < lim && *q != ' ') q++;
                size_t hn = (size_t)(q - b);
                if (hn > 0 && b[hn - 1] == '\r') hn--;          /* TrimSuffix "\r" */
                if (hn >= cap) hn = cap - 1;
                memcpy(host, b, hn); host[hn] = 0;
                break;
            }
        }
        if (e >= len) break;
        pos = e + 1;
    }
}

/* containsSQLKeywords — data.go:1624-1626 with the keyword alternation of data.go:123-126:
 * regexp.MatchString over strings.ToUpper(input) == "some keyword occurs as a substring".
 * ToUpper also maps U+0131 (dotless i) -> 'I' and U+017F (long s) -> 'S'; mirrored. */
static int contains_sql_keywords(const uint8_t* s, size_t n) {
    static const char* kw[] = { "SELECT", "INSERT INTO", "UPDATE", "DELETE FROM", "CREATE TABLE", "ALTER TABLE",
        "DROP TABLE", "TRUNCATE TABLE", "BEGIN", "COMMIT", "ROLLBACK", "SAVEPOINT", "CREATE INDEX", "DROP INDEX",
        "CREATE VIEW", "DROP VIEW", "GRANT", "REVOKE", "EXECUTE" };
    char* up = malloc(n + 1); size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t c = s[i];
        if (c == 0xC4 && i + 1 < n && s[i + 1] == 0xB1) { up[m++] = 'I'; i++; }
        else if (c == 0xC5 && i + 1 < n && s[i + 1] == 0xBF) { up[m++] = 'S'; i++; }
        else up[m++] = (c >= 'a' && c <= 'z') ? (char)(c - 32) : (char)c;
    }
    up[m] = 0;
    int found = 0;
    for (size_t k = 0; k < sizeof(kw) / sizeof(kw[0]) && !found; k++)
        if (memmem(up, m, kw[k], strlen(kw[k]))) found = 1;
    free(up);
    return found;
}

/* bytes.Split(b, {0}) helper: returns number of pieces, fills up to `want` (start,len) pairs */
static size_t split_nul(const uint8_t* b, size_t n, size_t* st, size_t* ln, size_t want) {
    size_t np = 0, s = 0;
    for (size_t i = 0; i <= n; i++) {
        if (i == n || b[i] == 0) { if (np < want) { st[np] = s; ln[np] = i - s; } np++; s = i + 1; }
    }
    return np;
}

static void copy_trunc(char* out, size_t cap, const void* src, size_t n) {
    if (cap == 0) return;
    if (n >= cap) n = cap - 1;
    memcpy(out, src, n); out[n] = 0;
}

/* parsePostgresCommand — aggregator/data.go:1474-1556.  Returns 0 ok (query in out), -1 error
 * (the caller drops the event, data.go:1328-1332). Statement key = "%d-%d-%s" of pid, fd, name
 * (getPgStmtKey :1619-1621 over getConnKey :910-912). */
int or_parse_postgres(oracle_t* o, uint32_t pid, uint64_t fd, const char* method,
                      const uint8_t* r, size_t n, char* out, size_t cap) {
    out[0] = 0;
    if (strcmp(method, "SIMPLE_QUERY") == 0) {
        if (n < 5) return -1;                                   /* "too short for a sql query" */
        r += 5; n -= 5;                                         /* skip 'Q' + 4 length bytes   */
        if (!contains_sql_keywords(r, n)) return -1;            /* "no sql command found"      */
        copy_trunc(out, cap, r, n);
        return 0;
    } else if (strcmp(method, "EXTENDED_QUERY") == 0) {
        if (n < 5) return -1;   /* Go would panic on r[0] / r[5:] for n<5; treated as a drop */
        uint8_t id = r[0];
        size_t st[3], ln[3];
        size_t np = split_nul(r + 5, n - 5, st, ln, 3);
        char key[512], name[256], query[1100];
        if (id == 'P') {
            if (np >= 3) { copy_trunc(name, sizeof name, r + 5 + st[0], ln[0]); copy_trunc(query, sizeof query, r + 5 + st[1], ln[1]); }
            else if (np == 2) {                                  /* query too long for the buffer */
                copy_trunc(name, sizeof name, r + 5 + st[0], ln[0]);
                copy_trunc(query, sizeof query - 3, r + 5 + st[1], ln[1]); strcat(query, "...");
            } else return -1;
            snprintf(key, sizeof key, "%u-%llu-%s", pid, (unsigned long long)fd, name);
            sm_put(&o->pg_stmts, key, query, 0);
            snprintf(out, cap, "PREPARE %s AS %s", name, query);
            return 0;
        } else if (id == 'B') {
            if (np >= 2) copy_trunc(name, sizeof name, r + 5 + st[1], ln[1]);
            else return -1;
            snprintf(key, sizeof key, "%u-%llu-%s", pid, (unsigned long long)fd, name);
            sm_ent* e = sm_find(&o->pg_stmts, key);
            if (!e || !e->sval || e->sval[0] == 0) { snprintf(out, cap, "EXECUTE %s *values*", name); return 0; }
            copy_trunc(out, cap, e->sval, strlen(e->sval));
            return 0;
        }
        return -1;                                              /* "could not parse extended query" */
    } else if (strcmp(method, "CLOSE_OR_TERMINATE") == 0) {
        copy_trunc(out, cap, r, n);
    }
    return 0;
}

/* parseMySQLCommand — aggregator/data.go:1431-1472. */
static int parse_mysql(oracle_t* o, uint32_t pid, uint64_t fd, uint32_t prep_id, const char* method,
                       const uint8_t* r, size_t n, char* out, size_t cap) {
    out[0] = 0;
    if (n < 5) return -1;
    r += 5; n -= 5;
    char key[96];
    if (strcmp(method, "TEXT_QUERY") == 0) {
        if (!contains_sql_keywords(r, n)) return -1;
    } else if (strcmp(method, "PREPARE_STMT") == 0) {
        char q[1100]; copy_trunc(q, sizeof q, r, n);
        snprintf(key, sizeof key, "%u-%llu-%u", pid, (unsigned long long)fd, prep_id);
        sm_put(&o->mysql_stmts, key, q, 0);
    } else if (strcmp(method, "EXEC_STMT") == 0 || strcmp(method, "STMT_CLOSE") == 0) {
        if (n < 4) return 0;    /* binary.LittleEndian.Uint32 panics (no recover) -> treat as passthrough */
        uint32_t sid = (uint32_t)r[0] | (uint32_t)r[1] << 8 | (uint32_t)r[2] << 16 | (uint32_t)r[3] << 24;
        snprintf(key, sizeof key, "%u-%llu-%u", pid, (unsigned long long)fd, sid);
        if (method[0] == 'E') {
            sm_ent* e = sm_find(&o->mysql_stmts, key);
            if (!e || !e->sval || e->sval[0] == 0) { snprintf(out, cap, "EXECUTE %u *values*", sid); return 0; }
            copy_trunc(out, cap, e->sval, strlen(e->sval));
            return 0;
        }
        sm_del(&o->mysql_stmts, key);
        snprintf(out, cap, "CLOSE STMT %u ", sid);
        return 0;
    }
    copy_trunc(out, cap, r, n);
    return 0;
}

/* parseMongoEvent — aggregator/data.go:1561-1617.  Slice-out-of-range panics are recovered by the
 * deferred recover() (:1562-1567), which leaves the unnamed results at their zero values
 * ("", nil): the event is then persisted with an empty path.  Returns 0 ok, -1 error.
 * Bounds are checked against the captured length.  Go reslices (`payload[:4]`, `payload[4:docLen]`) are legal up to the
 * slice's CAPACITY — the rest of the 1 KiB Payload array — so a capture whose declared lengths exceed the captured bytes
 * without filling the slot would be parsed by the reference over stale array bytes; that case is not modelled (it needs a
 * sender that lies about its lengths: a complete short message never has it, a truncated one fills the slot). */
static int parse_mongo(const uint8_t* p, size_t n, char* out, size_t cap) {
    out[0] = 0;
#define PANIC_IF(c) do { if (c) { out[0] = 0; return 0; } } while (0)
    PANIC_IF(n < 12); p += 12; n -= 12;
    PANIC_IF(n < 4);
    uint32_t opcode = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    PANIC_IF(n < 8); p += 8; n -= 8;
    if (opcode == 2012) { copy_trunc(out, cap, "compressed mongo event", 22); return 0; }
    if (opcode == 2013) {
        PANIC_IF(n < 1);
        uint8_t kind = p[0]; p += 1; n -= 1;
        if (kind == 0) {
            PANIC_IF(n < 4);
            uint32_t doc_len = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
            PANIC_IF(doc_len < 4 || doc_len > n);            /* payload[4:docLen] */
            p += 4; n = doc_len - 4;
            PANIC_IF(n < 1);
            if (p[0] != 2) return -1;                           /* "document element not a string" */
            p += 1; n -= 1;
            size_t el = 0; while (el < n && p[el] != 0) el++;
            PANIC_IF(el + 5 > n);
            uint32_t vlen = (uint32_t)p[el + 1] | (uint32_t)p[el + 2] << 8 | (uint32_t)p[el + 3] << 16 | (uint32_t)p[el + 4] << 24;
            const uint8_t* v = p + el + 5; size_t vn = n - (el + 5);
            PANIC_IF(vlen == 0 || vlen - 1 > vn);
            char elem[1100]; copy_trunc(elem, sizeof elem, p, el);
            char val[1100]; copy_trunc(val, sizeof val, v, vlen - 1);
            snprintf(out, cap, "%s %s", elem, val);
            return 0;
        }
    }
#undef PANIC_IF
    return -1;                                                  /* "could not parse mongo event" */
}

/* enum -> string tables — ebpf/l7_req/l7.go:47-72 (protocol) and :200-330 (methods). */
static const char* proto_str(uint8_t p) {
    static const char* t[] = { "UNKNOWN", "HTTP", "AMQP", "POSTGRES", "HTTP2", "REDIS", "KAFKA", "MYSQL", "MONGO" };
    return p <= 8 ? t[p] : "Unknown";
}
static const char* method_str(uint8_t proto, uint8_t m) {
    static const char* http[] = { "Unknown", "GET", "POST", "PUT", "PATCH", "DELETE", "HEAD", "CONNECT", "OPTIONS", "TRACE" };
    switch (proto) {
    case SG_PROTO_HTTP:     return (m >= 1 && m <= 9) ? http[m] : "Unknown";
    case SG_PROTO_AMQP:     return m == 1 ? "PUBLISH" : m == 2 ? "DELIVER" : "Unknown";
    case SG_PROTO_POSTGRES: return m == 1 ? "CLOSE_OR_TERMINATE" : m == 2 ? "SIMPLE_QUERY" : m == 3 ? "EXTENDED_QUERY" : "Unknown";
    case SG_PROTO_HTTP2:    return m == 1 ? "CLIENT_FRAME" : m == 2 ? "SERVER_FRAME" : "Unknown";
    case SG_PROTO_REDIS:    return m == 1 ? "COMMAND" : m == 2 ? "PUSHED_EVENT" : m == 3 ? "PING" : "Unknown";
    case SG_PROTO_KAFKA:    return m == 1 ? "PRODUCE_REQUEST" : m == 2 ? "FETCH_RESPONSE" : "Unknown";
    case SG_PROTO_MYSQL:    return m == 1 ? "TEXT_QUERY" : m == 2 ? "PREPARE_STMT" : m == 3 ? "EXEC_STMT" : m == 4 ? "STMT_CLOSE" : "Unknown";
    default:                return "Unknown";   /* MONGO / UNKNOWN: l7.go:730-734 */
    }
}

/* user-space view of one event, L7Event (l7.go:396-417) */
typedef struct {
    uint64_t fd, duration, write_time_ns; uint32_t pid, status, payload_size, prep_stmt_id;
    uint8_t protocol, method, tls; const uint8_t* payload;
    uint32_t saddr, daddr; uint16_t sport, dport; int16_t kafka_api_version;
} l7ev;

/* bpfL7Event layout — ebpf/l7_req/l7.go:345-369 (= struct l7_event, ebpf/c/l7.c:19-47):
 * fd@0 write_time_ns@8 pid@16 status@20 duration@24 protocol@32 method@33 payload@36
 * payload_size@1060 read_complete@1064 failed@1065 is_tls@1066 kafka_api_version@1068
 * prep_statement_id@1072 saddr@1076 sport@1080 daddr@1084 dport@1088; total 1096. */
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
static void decode_wire(const uint8_t* r, l7ev* e) {
    e->fd = rd64(r + 0); e->write_time_ns = rd64(r + 8); e->pid = rd32(r + 16); e->status = rd32(r + 20);
    e->duration = rd64(r + 24); e->protocol = r[32]; e->method = r[33]; e->payload = r + 36;
    e->payload_size = rd32(r + 1060); if (e->payload_size > 1024) e->payload_size = 1024;
    e->tls = r[1066] != 0; e->prep_stmt_id = rd32(r + 1072); e->kafka_api_version = (int16_t)(r[1068] | r[1069] << 8);
    e->saddr = rd32(r + 1076); e->sport = rd16(r + 1080); e->daddr = rd32(r + 1084); e->dport = rd16(r + 1088);
}

/* "error" classification of the aggregate: HTTP/HTTP2 >= 500; POSTGRES/REDIS/MYSQL == 2
 * (ERROR_RESPONSE ebpf/c/postgres.c:91, STATUS_ERROR redis.c:10, MYSQL_STATUS_FAILED mysql.c:36). */
static int is_error(uint8_t proto, uint32_t status) {
    if (proto == SG_PROTO_HTTP || proto == SG_PROTO_HTTP2) return status >= 500;
    if (proto == SG_PROTO_POSTGRES || proto == SG_PROTO_REDIS || proto == SG_PROTO_MYSQL) return status == 2;
    return 0;
}

typedef struct { char type[10]; char uid[OR_UID_MAX]; uint32_t ref; uint32_t obip; } endpoint;

/* setFromToV2 — aggregator/data.go:827-870 (getPodWithIP :812-817, getSvcWithIP :819-825).
 * Reverse DNS (getHostnameFromIP :1386-1405) is excluded from parity: treated as failing, so the
 * fallback ToUID = raw IP string applies (:862-863). Returns -1 when the source is not a pod. */
static int set_from_to_v2(oracle_t* o, const char* saddr, const char* daddr, uint32_t daddr_num,
                          const char* host_header, endpoint* from, endpoint* to) {
    sm_ent* p = sm_find(&o->pod_ip_to_uid, saddr);
    if (!p) return -1;                                          /* "error finding pod with sockets saddr" */
    strcpy(from->type, "pod"); copy_trunc(from->uid, OR_UID_MAX, p->sval, strlen(p->sval));
    from->ref = SG_MAKE_REF(SG_REF_KNOWN, sm_find(&o->uid_to_id, p->sval)->uval); from->obip = 0;

    sm_ent* s = sm_find(&o->svc_ip_to_uid, daddr);
    if (s) {
        strcpy(to->type, "service"); copy_trunc(to->uid, OR_UID_MAX, s->sval, strlen(s->sval));
        to->ref = SG_MAKE_REF(SG_REF_KNOWN, sm_find(&o->uid_to_id, s->sval)->uval); to->obip = 0;
    } else if ((p = sm_find(&o->pod_ip_to_uid, daddr)) != NULL) {
        strcpy(to->type, "pod"); copy_trunc(to->uid, OR_UID_MAX, p->sval, strlen(p->sval));
        to->ref = SG_MAKE_REF(SG_REF_KNOWN, sm_find(&o->uid_to_id, p->sval)->uval); to->obip = 0;
    } else {
        strcpy(to->type, "outbound");
        if (host_header && host_header[0]) {
            copy_trunc(to->uid, OR_UID_MAX, host_header, strlen(host_header));
            to->ref = SG_MAKE_REF(SG_REF_LABEL, intern_label(o, to->uid)); to->obip = 0;
        } else {
            copy_trunc(to->uid, OR_UID_MAX, daddr, strlen(daddr));
            to->ref = SG_MAKE_REF(SG_REF_OBIP, 0); to->obip = daddr_num;
        }
    }
    return 0;
}

/* f-3: per-edge log2 latency histogram and the percentiles read off it — include/servicegraph.h is the definition */
uint32_t or_hist_bin(uint64_t dur_ns) {
    if (dur_ns < (1ull << 17)) return 0;
    if (dur_ns >= (1ull << 31)) return SG_HIST_BINS - 1;
    uint32_t lg = 0; while ((dur_ns >> (lg + 1)) != 0) lg++;     /* floor(log2) */
    return lg - 16;
}
uint32_t or_percentile_us(const uint32_t* hist, uint32_t count, uint64_t max_ns, uint32_t q) {
    if (count == 0) return 0;
    uint64_t rank = ((uint64_t)count * q + 99) / 100; if (rank == 0) rank = 1;
    uint64_t cum = 0; uint32_t b = 0;
    for (; b < SG_HIST_BINS; b++) { cum += hist[b]; if (cum >= rank) break; }
    if (b >= SG_HIST_BINS) b = SG_HIST_BINS - 1;
    uint64_t edge = b == SG_HIST_BINS - 1 ? max_ns : (1ull << (17 + b));
    if (edge > max_ns) edge = max_ns;
    uint64_t us = edge / 1000ull;
    return us > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)us;
}
const uint32_t* or_edge_hist(const oracle_t* o) { return o->edge_hist; }

static w_edge* window_edge(oracle_t* o, const endpoint* f, const endpoint* t) {
    char key[2 * OR_UID_MAX + 32];
    snprintf(key, sizeof key, "%s\x1f%s\x1f%s\x1f%s", f->type, f->uid, t->type, t->uid);
    sm_ent* e = sm_find(&o->edge_index, key);
    if (e) return &o->wedges[e->uval];
    if (o->n_wedges == o->cap_wedges) {
        o->cap_wedges = o->cap_wedges ? o->cap_wedges * 2 : 1024;
        o->wedges = realloc(o->wedges, o->cap_wedges * sizeof(w_edge));
    }
    w_edge* w = &o->wedges[o->n_wedges];
    memset(w, 0, sizeof *w);
    w->from_ref = f->ref; w->to_ref = t->ref; w->from_obip = f->obip; w->to_obip = t->obip;
    sm_put(&o->edge_index, key, NULL, (uint32_t)o->n_wedges);
    o->n_wedges++;
    return w;
}

/* Everything after the payload parse, common to all process<Proto>Event handlers
 * (data.go:1081-1118, 1120-1160, 1208-1249, 1251-1285, 1287-1321, 1323-1362, 1035-1079):
 * build the DTO, join, optionally ReverseDirection (datastore/dto.go:226-231), "HTTPS" rewrite
 * (data.go:1240-1242), PersistRequest (datastore/backend.go:819-847). */
static int resolve_and_persist(oracle_t* o, uint32_t saddr, uint16_t sport, uint32_t daddr, uint16_t dport,
                               uint8_t proto, const char* protocol, const char* method, int tls,
                               uint32_t status, uint64_t duration, uint64_t write_ns,
                               const char* host_header, const char* path, int reverse, int is_kafka) {
    char sip[16], dip[16];
    or_int_to_ipv4(saddr, sip); or_int_to_ipv4(daddr, dip);      /* extractAddressPair */
    endpoint from, to;
    if (set_from_to_v2(o, sip, dip, daddr, host_header, &from, &to) != 0) { o->dropped_src++; return 0; }

    const char* fip = sip; const char* tip = dip; uint16_t fport = sport, tport = dport;
    if (reverse) {                                               /* ReverseDirection() */
        endpoint tmp = from; from = to; to = tmp;
        fip = dip; tip = sip; fport = dport; tport = sport;
    }
    int64_t st = start_time_ms(o, write_ns);

    if (o->log_n < o->log_limit) {
        if (o->log_n == o->log_cap) { o->log_cap = o->log_cap ? o->log_cap * 2 : 256; o->log = realloc(o->log, o->log_cap * sizeof(or_reqinfo)); }
        or_reqinfo* r = &o->log[o->log_n++];
        memset(r, 0, sizeof *r);
        r->start_time = st; r->latency = duration;
        strcpy(r->from_ip, fip); strcpy(r->from_type, from.type); strcpy(r->from_uid, from.uid); r->from_port = fport;
        strcpy(r->to_ip, tip); strcpy(r->to_type, to.type); strcpy(r->to_uid, to.uid); r->to_port = tport;
        const char* pr = protocol;
        if (proto == SG_PROTO_HTTP && tls) pr = "HTTPS";
        copy_trunc(r->protocol, sizeof r->protocol, pr, strlen(pr));
        r->status_code = status; copy_trunc(r->method, sizeof r->method, method, strlen(method));
        copy_trunc(r->path, sizeof r->path, path ? path : "", path ? strlen(path) : 0);
        r->tls = (uint8_t)tls; r->is_kafka = (uint8_t)is_kafka;
    }
    o->persisted++;

    /* Part 2 hook: integer accumulation per edge */
    w_edge* w = window_edge(o, &from, &to);
    w->count += 1;
    w->err += (uint32_t)is_error(proto, status > 0xFFFF ? 0xFFFF : status);
    w->sum_ns += duration;
    if (duration > w->max_ns) w->max_ns = duration;
    uint64_t us = duration / 1000ull;
    w->sumsq_us += us * us;
    w->hist[or_hist_bin(duration)] += 1;
    if (st < o->tmin) o->tmin = st;
    if (st > o->tmax) o->tmax = st;
    o->wevents++;
    return 1;
}

/* sendOpenConnection's join + PersistAliveConnection — aggregator/data.go:1641-1677: the source
 * must be a pod (else the connection is ignored, not counted); destination service first, then pod,
 * else ("outbound", Daddr).  No Host header, no reverse DNS, no direction reversal.
 * Part 2 hook: the edge is created if needed and its alive count grows by one. */
static int persist_alive(oracle_t* o, uint32_t saddr, uint16_t sport, uint32_t daddr, uint16_t dport, int64_t check_ms) {
    char sip[16], dip[16];
    or_int_to_ipv4(saddr, sip); or_int_to_ipv4(daddr, dip);
    endpoint from, to;
    if (set_from_to_v2(o, sip, dip, daddr, "", &from, &to) != 0) return 0;
    if (o->alive_n < o->log_limit) {
        if (o->alive_n == o->alive_cap) { o->alive_cap = o->alive_cap ? o->alive_cap * 2 : 64; o->alive_log = realloc(o->alive_log, o->alive_cap * sizeof(or_alive)); }
        or_alive* a = &o->alive_log[o->alive_n++];
        memset(a, 0, sizeof *a);
        a->check_time = check_ms;
        strcpy(a->from_ip, sip); strcpy(a->from_type, "pod"); strcpy(a->from_uid, from.uid); a->from_port = sport;
        strcpy(a->to_ip, dip); strcpy(a->to_type, to.type); strcpy(a->to_uid, to.uid); a->to_port = dport;
    }
    o->alive_persisted++;
    window_edge(o, &from, &to)->alive += 1;
    return 1;
}

static uint32_t ipv4_to_int(const char* s) {
    unsigned a = 0, b = 0, c = 0, d = 0;
    if (sscanf(s, "%u.%u.%u.%u", &a, &b, &c, &d) != 4) return 0;
    return (a << 24) | (b << 16) | (c << 8) | d;
}

or_sockline* or_sockline_of(oracle_t* o, uint32_t pid, uint64_t fd) {
    char key[48]; snprintf(key, sizeof key, "%u:%llu", pid, (unsigned long long)fd);
    sm_ent* e = sm_find(&o->sock_index, key);
    return e ? o->socklines[e->uval] : NULL;
}
size_t or_sockline_count(const oracle_t* o) { size_t n = 0; for (size_t i = 0; i < o->n_socklines; i++) n += o->socklines[i] != NULL; return n; }

/* "/proc" for NewSocketLine's fetch (NULL: lines start empty) and the time.Now() it stamps the seeded value with */
void or_set_proc_root(oracle_t* o, const char* root, uint64_t now_user_ns) {
    free(o->proc_root); o->proc_root = root ? strdup(root) : NULL; o->now_user_ns = now_user_ns;
}

/* processExit — aggregator/data.go:363-398: clearProc (cluster.go:97-110: the process' socket map, i.e. every line
 * of the pid, is gone), the pid's HTTP/2 parsers and Postgres statements by STRING PREFIX of the decimal pid
 * (pid 12 also clears 123's), rate limiter (not on this path).  The MySQL loop :391-397 ranges over pgStmts —
 * whose keys with this prefix were deleted just above — so no MySQL statement is ever removed here. */
void or_process_exit(oracle_t* o, uint32_t pid) {
    for (size_t i = 0; i < o->n_socklines; i++) {
        or_sockline* sl = o->socklines[i];
        uint64_t fd;
        if (!sl || or_sl_owner(sl, &fd) != pid) continue;
        char key[48]; snprintf(key, sizeof key, "%u:%llu", pid, (unsigned long long)fd);
        sm_del(&o->sock_index, key);
        or_sl_destroy(sl); o->socklines[i] = NULL;
    }
    or_h2_proc_exit(o->h2, pid);
    char pfx[16]; snprintf(pfx, sizeof pfx, "%u", pid);
    sm_del_prefix(&o->pg_stmts, pfx);
}
size_t or_pg_stmt_count(const oracle_t* o) { return o->pg_stmts.used; }

/* processTcpConnect — aggregator/data.go:404-506 */
int or_process_tcp(oracle_t* o, uint32_t type, uint32_t pid, uint64_t fd, uint64_t ts,
                   const char* saddr, uint16_t sport, const char* daddr, uint16_t dport) {
    if (type != OR_TCP_ESTABLISHED && type != OR_TCP_CLOSED) return 0;
    if (strcmp(saddr, "127.0.0.1") == 0 || strcmp(daddr, "127.0.0.1") == 0) return 0;   /* :409-411, :456-458 */
    or_sockline* sl = or_sockline_of(o, pid, fd);
    if (type == OR_TCP_ESTABLISHED) {
        if (!sl) {                                   /* :416-438: signalled for creation, event re-queued until it exists */
            if (o->n_socklines == o->cap_socklines) { o->cap_socklines = o->cap_socklines ? o->cap_socklines * 2 : 64; o->socklines = realloc(o->socklines, o->cap_socklines * sizeof(or_sockline*)); }
            sl = or_sl_create(pid, fd);
            char key[48]; snprintf(key, sizeof key, "%u:%llu", pid, (unsigned long long)fd);
            sm_put(&o->sock_index, key, NULL, (uint32_t)o->n_socklines);
            o->socklines[o->n_socklines++] = sl;
            /* sock_num_line.go:38-54: populated from /proc BEFORE the re-queued event finds the line */
            if (o->proc_root) (void)or_sl_seed_from_proc(sl, o->proc_root, o->first_kernel - (o->first_user - o->now_user_ns));   /* data.go:1745-1747 */
        }
        or_sockinfo si; memset(&si, 0, sizeof si);
        si.pid = pid; si.fd = fd; si.sport = sport; si.dport = dport;
        copy_trunc(si.saddr, sizeof si.saddr, saddr, strlen(saddr)); copy_trunc(si.daddr, sizeof si.daddr, daddr, strlen(daddr));
        or_sl_add(sl, ts, &si);                      /* :440-450 */
        return 1;
    }
    if (!sl) return 0;                               /* CLOSED without a line: dropped (:472-477) */
    or_sl_add(sl, ts, NULL);                         /* :480-483 */
    or_h2_conn_closed(o->h2, pid, fd);               /* :485-494 */
    { char ck[48]; snprintf(ck, sizeof ck, "%u-%llu", pid, (unsigned long long)fd);      /* :496-503: every pgStmts key that STARTS WITH */
      sm_del_prefix(&o->pg_stmts, ck); }                                                  /* "pid-fd" goes - also "pid-fd7-..." of another fd   */
    return 1;
}

size_t or_process_tcp_wire(oracle_t* o, const uint8_t* recs, size_t n) {
    size_t added = 0;
    for (size_t i = 0; i < n; i++) {
        const uint8_t* r = recs + i * OR_TCP_WIRE_SIZE;
        char s[16], d[16];
        snprintf(s, sizeof s, "%u.%u.%u.%u", r[28], r[29], r[30], r[31]);    /* tcp.go:241-242 */
        snprintf(d, sizeof d, "%u.%u.%u.%u", r[44], r[45], r[46], r[47]);
        added += (size_t)or_process_tcp(o, rd32(r + 16), rd32(r + 20), rd64(r), rd64(r + 8), s, rd16(r + 24), d, rd16(r + 26));
    }
    return added;
}

/* clearSocketLines, one tick — aggregator/data.go:1681-1716 (sendOpenConnection :1628-1679) */
size_t or_sweep_socket_lines(oracle_t* o, int64_t now_ms, int send_alive) {
    size_t sent = 0;
    for (size_t i = 0; i < o->n_socklines; i++) {
        or_sockline* sl = o->socklines[i];
        if (!sl) continue;
        size_t len = or_sl_len(sl);
        if (send_alive && len > 0) {
            or_sockinfo si; uint64_t ts, lm;
            if (or_sl_at(sl, len - 1, &ts, &lm, &si) == 1)           /* last value is an open socket */
                sent += (size_t)persist_alive(o, ipv4_to_int(si.saddr), si.sport, ipv4_to_int(si.daddr), si.dport, now_ms);
        }
        or_sl_delete_unused(sl);
    }
    return sent;
}
size_t or_alive_count(const oracle_t* o) { return o->alive_persisted; }
const or_alive* or_alive_at(const oracle_t* o, size_t i) { return i < o->alive_n ? &o->alive_log[i] : NULL; }

/* processL7 — aggregator/data.go:1364-1383 — and the per-protocol handlers it dispatches to. */
static size_t process_one(oracle_t* o, const l7ev* d, uint32_t kafka_msgs) {
    const char* protocol = proto_str(d->protocol);
    const char* method = method_str(d->protocol, d->method);
    char host[OR_UID_MAX], path[OR_HTTP_PATH_CAP], m[OR_HTTP_TOK_CAP], v[OR_HTTP_TOK_CAP];
    host[0] = path[0] = 0;
    switch (d->protocol) {
    case SG_PROTO_HTTP:                                          /* processHttpEvent :1208-1249 */
        or_parse_http_payload((const char*)d->payload, d->payload_size, m, path, v, host);
        return (size_t)resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol, method, d->tls,
                                           d->status, d->duration, d->write_time_ns, host, path, 0, 0);
    case SG_PROTO_POSTGRES:                                      /* processPostgresEvent :1323-1362 */
        if (or_parse_postgres(o, d->pid, d->fd, method, d->payload, d->payload_size, path, sizeof path) != 0) { o->dropped_parse++; return 0; }
        return (size_t)resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol, method, d->tls,
                                           d->status, d->duration, d->write_time_ns, "", path, 0, 0);
    case SG_PROTO_MYSQL:                                         /* processMySQLEvent :1287-1321 */
        if (parse_mysql(o, d->pid, d->fd, d->prep_stmt_id, method, d->payload, d->payload_size, path, sizeof path) != 0) { o->dropped_parse++; return 0; }
        return (size_t)resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol, method, d->tls,
                                           d->status, d->duration, d->write_time_ns, "", path, 0, 0);
    case SG_PROTO_MONGO:                                         /* processMongoEvent :1251-1285 */
        if (parse_mongo(d->payload, d->payload_size, path, sizeof path) != 0) { o->dropped_parse++; return 0; }
        return (size_t)resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol, method, d->tls,
                                           d->status, d->duration, d->write_time_ns, "", path, 0, 0);
    case SG_PROTO_REDIS:                                         /* processRedisEvent :1120-1160 */
        copy_trunc(path, sizeof path, d->payload, d->payload_size);
        return (size_t)resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol, method, d->tls,
                                           d->status, d->duration, d->write_time_ns, "", path,
                                           strcmp(method, "PUSHED_EVENT") == 0, 0);
    case SG_PROTO_AMQP:                                          /* processAmqpEvent :1081-1118 */
        return (size_t)resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol, method, d->tls,
                                           d->status, d->duration, d->write_time_ns, "", "",
                                           strcmp(method, "DELIVER") == 0, 0);
    case SG_PROTO_KAFKA: {                                       /* processKafkaEvent :1035-1079 */
        /* decodeKafkaPayload (:929-1017) is out of scope; its message count is a side input.
         * 0 messages => event dropped (:1037-1039); one KafkaEvent per message; the first failing
         * setFromToV2 returns from the whole handler (:1065-1068). */
        size_t done = 0;
        if (o->kafka_decode) {                                   /* decodeKafkaPayload :929-1017 */
            or_kafka_result* kr = or_kafka_decode(d->payload, d->payload_size, d->method, d->kafka_api_version);
            const size_t n = or_kafka_status(kr) == 0 ? or_kafka_count(kr) : 0;
            if (n == 0) o->dropped_parse++;                      /* err != nil || len(kafkaMessages) == 0 (:1037-1039) */
            for (size_t k = 0; k < n; k++) {
                const uint8_t *t, *kk, *vv; size_t tn, kn, vn; int32_t part; char topic[OR_HTTP_PATH_CAP];
                or_kafka_msg(kr, k, &t, &tn, &part, &kk, &kn, &vv, &vn);
                copy_trunc(topic, sizeof topic, t, tn);
                if (!resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol, d->method == 2 ? "CONSUME" : "PUBLISH",
                                         d->tls, d->status, d->duration, d->write_time_ns, "", topic, 0, 1)) break;
                done++;
            }
            or_kafka_result_free(kr);
            return done;
        }
        for (uint32_t k = 0; k < kafka_msgs; k++) {
            int ok = resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, protocol,
                                         d->method == 2 ? "CONSUME" : "PUBLISH", d->tls, d->status, d->duration,
                                         d->write_time_ns, "", "", 0, 1);
            if (!ok) break;
            done++;
        }
        return done;
    }
    case SG_PROTO_HTTP2: {                                       /* processHttp2Event :1019-1033 -> processHttp2Frames :544-810 */
        or_h2_out r;
        if (!or_h2_event(o->h2, d->pid, d->fd, d->method, d->payload, d->payload_size, d->write_time_ns, d->tls, &r)) return 0;
        return (size_t)resolve_and_persist(o, d->saddr, d->sport, d->daddr, d->dport, d->protocol, r.protocol, r.method, d->tls,
                                           r.status_code, r.latency, d->write_time_ns, r.authority, r.path, 0, 0);
    }
    default:   /* UNKNOWN ignored */
        return 0;
    }
}

size_t or_process_l7_wire(oracle_t* o, const uint8_t* recs, size_t n, const uint32_t* kafka_msgs) {
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        l7ev d; decode_wire(recs + i * OR_L7_WIRE_SIZE, &d);
        total += process_one(o, &d, kafka_msgs ? kafka_msgs[i] : 1u);
    }
    return total;
}

size_t or_process_packed(oracle_t* o, const sg_event* ev, size_t n, const char* const* labels, size_t n_labels) {
    size_t total = 0;
    /* the packer owns the label ids: host_label i+1 <-> labels[i]; register them in table order so
     * that the oracle's LABEL numbering is the packer's (and the label count is the table size) */
    for (size_t i = 0; i < n_labels; i++) {
        uint32_t id = intern_label(o, labels[i]);
        (void)id;
    }
    for (size_t i = 0; i < n; i++) {
        const sg_event* e = &ev[i];
        if (e->flags & SG_EV_ALIVE) { persist_alive(o, e->saddr, 0, e->daddr, 0, 0); continue; }
        const char* host = "";
        if (e->host_label != 0 && e->host_label <= n_labels) host = labels[e->host_label - 1];
        const char* method = e->protocol == SG_PROTO_KAFKA ? ((e->flags & SG_EV_CONSUME) ? "CONSUME" : "PUBLISH") : "";
        total += (size_t)resolve_and_persist(o, e->saddr, 0, e->daddr, 0, e->protocol, proto_str(e->protocol), method,
                                             (e->flags & SG_EV_TLS) != 0, e->status, e->duration_ns, e->write_time_ns,
                                             host, "", (e->flags & SG_EV_REVERSE) != 0, e->protocol == SG_PROTO_KAFKA);
    }
    return total;
}

size_t or_reqinfo_count(const oracle_t* o) { return o->persisted; }
size_t or_reqinfo_logged(const oracle_t* o) { return o->log_n; }
const or_reqinfo* or_reqinfo_at(const oracle_t* o, size_t i) { return i < o->log_n ? &o->log[i] : NULL; }
uint64_t or_dropped_src(const oracle_t* o) { return o->dropped_src; }
uint64_t or_dropped_parse(const oracle_t* o) { return o->dropped_parse; }
size_t or_label_count(const oracle_t* o) { return o->n_labels; }
const char* or_label_at(const oracle_t* o, size_t i) { return i < o->n_labels ? o->labels[i] : NULL; }
size_t or_known_count(const oracle_t* o) { return o->n_known; }

/* ------------------------------------------------------------------------------------------ */
/* Part 2 — edge aggregation + GraphSAGE scoring (defined here; no reference counterpart)     */
/* ------------------------------------------------------------------------------------------ */

uint32_t or_hash32(uint32_t h) {    /* murmur3 fmix32 */
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}

static size_t layer_in(uint32_t l) { return l == 0 ? SG_F_IN : SG_F_HID; }

size_t or_weights_count(uint32_t layers) {
    size_t n = 0;
    for (uint32_t l = 0; l < layers; l++) n += 2 * layer_in(l) * SG_F_HID + SG_F_HID;
    n += 2 * SG_F_HID * SG_F_HID + SG_F_EDGE * SG_F_HID + SG_F_HID + SG_F_HID + 1;
    return n;
}

typedef struct { uint32_t from, to; size_t src; } sort_edge;
static int cmp_sort_edge(const void* a, const void* b) {
    const sort_edge* x = a; const sort_edge* y = b;
    if (x->from != y->from) return x->from < y->from ? -1 : 1;
    if (x->to != y->to) return x->to < y->to ? -1 : 1;
    return 0;
}
static int cmp_u32(const void* a, const void* b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }

static size_t lower_bound_u32(const uint32_t* a, size_t n, uint32_t v) {
    size_t lo = 0, hi = n; while (lo < hi) { size_t m = (lo + hi) / 2; if (a[m] < v) lo = m + 1; else hi = m; } return lo;
}

#define ST_OUT_DEG 0
#define ST_IN_DEG 1
#define ST_OUT_CNT 2
#define ST_IN_CNT 3
#define ST_OUT_ERR 4
#define ST_IN_ERR 5
#define ST_OUT_SUM 6
#define ST_IN_SUM 7
#define ST_OUT_SSQ 8
#define ST_IN_SSQ 9
#define ST_OUT_ALIVE 10
#define ST_IN_ALIVE 11

static double mean_us(uint64_t sum_ns, uint64_t cnt) { return cnt ? ((double)sum_ns / 1000.0) / (double)cnt : 0.0; }
static double std_us(uint64_t sum_ns, uint64_t ssq_us, uint64_t cnt) {
    if (!cnt) return 0.0;
    double m = mean_us(sum_ns, cnt);
    double v = (double)ssq_us / (double)cnt - m * m;
    return v > 0.0 ? sqrt(v) : 0.0;
}

/* node features x_v — DESIGN.md §features.  All via fp64, one final rounding to fp32. */
static void node_features(const uint64_t* s, const uint64_t* mx, uint8_t kind, float* x) {
    memset(x, 0, SG_F_IN * sizeof(float));
    uint64_t oc = s[ST_OUT_CNT], ic = s[ST_IN_CNT];
    x[0] = (float)log1p((double)s[ST_OUT_DEG]);
    x[1] = (float)log1p((double)s[ST_IN_DEG]);
    x[2] = (float)log1p((double)oc);
    x[3] = (float)log1p((double)ic);
    x[4] = (float)log1p(mean_us(s[ST_OUT_SUM], oc) / 1000.0);
    x[5] = (float)log1p(mean_us(s[ST_IN_SUM], ic) / 1000.0);
    x[6] = oc ? (float)((double)s[ST_OUT_ERR] / (double)oc) : 0.0f;
    x[7] = ic ? (float)((double)s[ST_IN_ERR] / (double)ic) : 0.0f;
    x[8] = (float)log1p((double)mx[0] / 1e6);
    x[9] = (float)log1p((double)mx[1] / 1e6);
    x[10] = kind == SG_NODE_POD ? 1.0f : 0.0f;
    x[11] = kind == SG_NODE_SERVICE ? 1.0f : 0.0f;
    x[12] = kind == 0 ? 1.0f : 0.0f;                            /* outbound */
    x[13] = (float)log1p(std_us(s[ST_OUT_SUM], s[ST_OUT_SSQ], oc) / 1000.0);
    x[14] = (float)log1p(std_us(s[ST_IN_SUM], s[ST_IN_SSQ], ic) / 1000.0);
    x[15] = 1.0f;
    x[16] = (float)log1p((double)s[ST_OUT_ALIVE]);              /* open connections from / to the node (f-2) */
    x[17] = (float)log1p((double)s[ST_IN_ALIVE]);
}

#define SG_MEAN_SLOTS 16
#define SG_MEAN_BLOCK 512   /* neighbours per block of the canonical mean (a multiple of SG_MEAN_SLOTS) */

size_t or_window_close(oracle_t* o, const float* W, uint32_t L) {
    free_closed(o);
    size_t E = o->n_wedges;

    /* 1. canonical node numbering: KNOWN ids, then LABEL ids, then OBIP by ascending IP */
    uint32_t* ob = malloc((2 * E + 1) * sizeof(uint32_t)); size_t nob = 0;
    for (size_t i = 0; i < E; i++) {
        if (SG_REF_TYPE(o->wedges[i].from_ref) == SG_REF_OBIP) ob[nob++] = o->wedges[i].from_obip;
        if (SG_REF_TYPE(o->wedges[i].to_ref) == SG_REF_OBIP) ob[nob++] = o->wedges[i].to_obip;
    }
    qsort(ob, nob, sizeof(uint32_t), cmp_u32);
    size_t u = 0; for (size_t i = 0; i < nob; i++) if (i == 0 || ob[i] != ob[i - 1]) ob[u++] = ob[i];
    nob = u;
    o->obips = ob; o->n_obips = nob;
    size_t NK = o->n_known, NL = o->n_labels, N = NK + NL + nob;
    o->n_nodes = N;

    sort_edge* se = malloc((E + 1) * sizeof(sort_edge));
    for (size_t i = 0; i < E; i++) {
        const w_edge* w = &o->wedges[i];
        uint32_t r[2] = { w->from_ref, w->to_ref }, ip[2] = { w->from_obip, w->to_obip }, d[2];
        for (int k = 0; k < 2; k++) {
            uint32_t t = SG_REF_TYPE(r[k]), v = SG_REF_VALUE(r[k]);
            d[k] = t == SG_REF_KNOWN ? v : t == SG_REF_LABEL ? (uint32_t)(NK + v) : (uint32_t)(NK + NL + lower_bound_u32(ob, nob, ip[k]));
        }
        se[i].from = d[0]; se[i].to = d[1]; se[i].src = i;
    }
    qsort(se, E, sizeof(sort_edge), cmp_sort_edge);

    /* 2. CSR + integer node stats */
    uint32_t* rowptr = calloc(N + 1, sizeof(uint32_t));
    for (size_t i = 0; i < E; i++) rowptr[se[i].from + 1]++;
    for (size_t v = 0; v < N; v++) rowptr[v + 1] += rowptr[v];
    uint64_t* ss = calloc(N * SG_NODE_STAT_SUM_WORDS + 1, sizeof(uint64_t));
    uint64_t* sm = calloc(N * SG_NODE_STAT_MAX_WORDS + 1, sizeof(uint64_t));
    for (size_t i = 0; i < E; i++) {
        const w_edge* w = &o->wedges[se[i].src];
        uint64_t* a = ss + (size_t)se[i].from * SG_NODE_STAT_SUM_WORDS; uint64_t* b = ss + (size_t)se[i].to * SG_NODE_STAT_SUM_WORDS;
        a[ST_OUT_DEG]++; b[ST_IN_DEG]++;
        a[ST_OUT_CNT] += w->count; b[ST_IN_CNT] += w->count;
        a[ST_OUT_ERR] += w->err; b[ST_IN_ERR] += w->err;
        a[ST_OUT_SUM] += w->sum_ns; b[ST_IN_SUM] += w->sum_ns;
        a[ST_OUT_SSQ] += w->sumsq_us; b[ST_IN_SSQ] += w->sumsq_us;
        a[ST_OUT_ALIVE] += w->alive; b[ST_IN_ALIVE] += w->alive;
        uint64_t* ma = sm + (size_t)se[i].from * 2; uint64_t* mb = sm + (size_t)se[i].to * 2;
        if (w->max_ns > ma[0]) ma[0] = w->max_ns;
        if (w->max_ns > mb[1]) mb[1] = w->max_ns;
    }
    o->st_sum = ss; o->st_max = sm;

    /* 3. node features */
    float* x0 = calloc(N * SG_F_IN + 1, sizeof(float));
    for (size_t v = 0; v < N; v++) {
        uint8_t kind = v < NK ? o->id_kind[v] : 0;
        node_features(ss + v * SG_NODE_STAT_SUM_WORDS, sm + v * 2, kind, x0 + v * SG_F_IN);
    }
    o->x0 = x0;

    /* 4. L GraphSAGE-mean layers.
     *    mean: 16 interleaved partial sums over the ascending neighbour list, combined in slot
     *    order, divided by deg.  dense: fmaf chain, acc = b; k over self features, then mean. */
    const float* wp = W;
    const float* hin = x0;
    for (uint32_t l = 0; l < L; l++) {
        size_t Fi = layer_in(l);
        const float* Ws = wp; const float* Wn = Ws + Fi * SG_F_HID; const float* b = Wn + Fi * SG_F_HID;
        wp = b + SG_F_HID;
        float* hout = calloc(N * SG_F_HID + 1, sizeof(float));
        float mean[SG_F_HID], part[SG_MEAN_SLOTS][SG_F_HID];
        for (size_t v = 0; v < N; v++) {
            uint32_t beg = rowptr[v], end = rowptr[v + 1], deg = end - beg;
            for (size_t k = 0; k < Fi; k++) mean[k] = 0.0f;
            if (deg) {
                /* blocks of SG_MEAN_BLOCK neighbours: inside a block 16 interleaved slot sums (neighbour i -> slot
                 * i % 16, ascending i), combined in slot order; block sums added in block order */
                float total[SG_F_HID];
                for (uint32_t b0 = 0; b0 < deg; b0 += SG_MEAN_BLOCK) {
                    uint32_t b1 = b0 + SG_MEAN_BLOCK < deg ? b0 + SG_MEAN_BLOCK : deg;
                    for (int s = 0; s < SG_MEAN_SLOTS; s++) for (size_t k = 0; k < Fi; k++) part[s][k] = 0.0f;
                    for (uint32_t i = b0; i < b1; i++) {
                        const float* hu = hin + (size_t)se[beg + i].to * Fi;
                        float* p = part[i % SG_MEAN_SLOTS];
                        for (size_t k = 0; k < Fi; k++) p[k] = p[k] + hu[k];
                    }
                    for (size_t k = 0; k < Fi; k++) {
                        float t = part[0][k];
                        for (int s = 1; s < SG_MEAN_SLOTS; s++) t = t + part[s][k];
                        total[k] = b0 == 0 ? t : total[k] + t;
                    }
                }
                for (size_t k = 0; k < Fi; k++) mean[k] = total[k] / (float)deg;
            }
            const float* hv = hin + v * Fi;
            for (size_t j = 0; j < SG_F_HID; j++) {
                float acc = b[j];
                for (size_t k = 0; k < Fi; k++) acc = fmaf(hv[k], Ws[k * SG_F_HID + j], acc);
                for (size_t k = 0; k < Fi; k++) acc = fmaf(mean[k], Wn[k * SG_F_HID + j], acc);
                hout[v * SG_F_HID + j] = acc > 0.0f ? acc : 0.0f;
            }
        }
        o->h[l + 1] = hout;
        hin = hout;
    }

    /* 5. score head: P = b1 + h Wu, Q = h Wv per node; per edge t = P[u]+Q[v] + e We; ReLU; w2; sigmoid */
    const float* Wu = wp; const float* Wv = Wu + SG_F_HID * SG_F_HID; const float* We = Wv + SG_F_HID * SG_F_HID;
    const float* b1 = We + SG_F_EDGE * SG_F_HID; const float* w2 = b1 + SG_F_HID; const float* b2 = w2 + SG_F_HID;
    size_t Fl = L == 0 ? SG_F_IN : SG_F_HID;   /* L>=1 always in practice */
    float* P = calloc(N * SG_F_HID + 1, sizeof(float)); float* Q = calloc(N * SG_F_HID + 1, sizeof(float));
    for (size_t v = 0; v < N; v++) {
        const float* hv = hin + v * Fl;
        for (size_t j = 0; j < SG_F_HID; j++) {
            float p = b1[j], q = 0.0f;
            for (size_t k = 0; k < SG_F_HID && k < Fl; k++) { p = fmaf(hv[k], Wu[k * SG_F_HID + j], p); q = fmaf(hv[k], Wv[k * SG_F_HID + j], q); }
            P[v * SG_F_HID + j] = p; Q[v * SG_F_HID + j] = q;
        }
    }

    o->edges = calloc(E + 1, sizeof(or_edge)); o->n_edges = E;
    free(o->edge_hist); o->edge_hist = calloc((E + 1) * SG_HIST_BINS, sizeof(uint32_t));
    for (size_t i = 0; i < E; i++) {
        const w_edge* w = &o->wedges[se[i].src];
        or_edge* oe = &o->edges[i];
        uint32_t uu = se[i].from, vv = se[i].to;
        /* refs as the engine reports them: OBIP payload = index into the ascending ip list */
        uint32_t fr = w->from_ref, tr = w->to_ref;
        if (SG_REF_TYPE(fr) == SG_REF_OBIP) fr = SG_MAKE_REF(SG_REF_OBIP, lower_bound_u32(ob, nob, w->from_obip));
        if (SG_REF_TYPE(tr) == SG_REF_OBIP) tr = SG_MAKE_REF(SG_REF_OBIP, lower_bound_u32(ob, nob, w->to_obip));
        oe->row.from_ref = fr; oe->row.to_ref = tr;
        oe->row.count = w->count; oe->row.err_count = w->err;
        oe->row.sum_ns = w->sum_ns; oe->row.max_ns = w->max_ns; oe->row.sumsq_us = w->sumsq_us;
        oe->row.alive = w->alive;
        memcpy(o->edge_hist + i * SG_HIST_BINS, w->hist, sizeof w->hist);
        oe->row.p50_us = or_percentile_us(w->hist, w->count, w->max_ns, 50);
        oe->row.p99_us = or_percentile_us(w->hist, w->count, w->max_ns, 99);

        /* edge features */
        const uint64_t* su = ss + (size_t)uu * SG_NODE_STAT_SUM_WORDS;
        double m_e = mean_us(w->sum_ns, w->count);
        double s_e = std_us(w->sum_ns, w->sumsq_us, w->count);
        double mu_src = mean_us(su[ST_OUT_SUM], su[ST_OUT_CNT]);
        double sd_src = std_us(su[ST_OUT_SUM], su[ST_OUT_SSQ], su[ST_OUT_CNT]);
        double z = (m_e - mu_src) / (sd_src > 1.0 ? sd_src : 1.0);
        float lat_z = (float)z;
        float err_ratio = w->count ? (float)((double)w->err / (double)w->count) : 0.0f;
        float zc = lat_z < -8.0f ? -8.0f : (lat_z > 8.0f ? 8.0f : lat_z);
        float e[SG_F_EDGE];
        e[0] = (float)log1p((double)w->count);
        e[1] = (float)log1p(m_e / 1000.0);
        e[2] = (float)log1p(s_e / 1000.0);
        e[3] = (float)log1p((double)w->max_ns / 1e6);
        e[4] = err_ratio;
        e[5] = (float)log1p((double)w->err);
        e[6] = zc * 0.125f;
        e[7] = 1.0f;

        float r[SG_F_HID];
        for (size_t j = 0; j < SG_F_HID; j++) {
            float t = P[(size_t)uu * SG_F_HID + j] + Q[(size_t)vv * SG_F_HID + j];
            for (size_t k = 0; k < SG_F_EDGE; k++) t = fmaf(e[k], We[k * SG_F_HID + j], t);
            t = t > 0.0f ? t : 0.0f;
            r[j] = t * w2[j];
        }
        /* butterfly tree over the 64 lanes: strides 32,16,8,4,2,1 */
        for (size_t s = SG_F_HID / 2; s >= 1; s >>= 1) {
            float nr[SG_F_HID];
            for (size_t j = 0; j < SG_F_HID; j++) nr[j] = r[j] + r[j ^ s];
            memcpy(r, nr, sizeof nr);
        }
        float logit = r[0] + b2[0];
        oe->row.score = 1.0f / (1.0f + expf(-logit));
        oe->row.lat_z = lat_z;
        oe->row.err_ratio = err_ratio;

        /* strings for the test harness */
        const uint32_t refs[2] = { w->from_ref, w->to_ref }; const uint32_t ips[2] = { w->from_obip, w->to_obip };
        char* types[2] = { oe->from_type, oe->to_type }; char* uids[2] = { oe->from_uid, oe->to_uid };
        for (int k = 0; k < 2; k++) {
            uint32_t t = SG_REF_TYPE(refs[k]), val = SG_REF_VALUE(refs[k]);
            if (t == SG_REF_KNOWN) { strcpy(types[k], o->id_kind[val] == SG_NODE_SERVICE ? "service" : "pod"); copy_trunc(uids[k], OR_UID_MAX, o->id_to_uid[val], strlen(o->id_to_uid[val])); }
            else if (t == SG_REF_LABEL) { strcpy(types[k], "outbound"); copy_trunc(uids[k], OR_UID_MAX, o->labels[val], strlen(o->labels[val])); }
            else { strcpy(types[k], "outbound"); or_int_to_ipv4(ips[k], uids[k]); }
        }
    }
    free(P); free(Q); free(rowptr); free(se);

    o->ctmin = o->tmin; o->ctmax = o->tmax; o->cevents = o->wevents;
    /* reset the open window (tables, interning and prepared statements persist) */
    sm_free(&o->edge_index); sm_init(&o->edge_index);
    o->n_wedges = 0; o->tmin = INT64_MAX; o->tmax = INT64_MIN; o->wevents = 0;
    return E;
}

size_t or_edge_count(const oracle_t* o) { return o->n_edges; }
const or_edge* or_edge_at(const oracle_t* o, size_t i) { return i < o->n_edges ? &o->edges[i] : NULL; }
size_t or_node_count(const oracle_t* o) { return o->n_nodes; }
const float* or_node_features(const oracle_t* o) { return o->x0; }
const float* or_layer_output(const oracle_t* o, uint32_t l) { return l <= SG_MAX_LAYERS ? o->h[l] : NULL; }
const uint64_t* or_node_stats_sum(const oracle_t* o) { return o->st_sum; }
const uint64_t* or_node_stats_max(const oracle_t* o) { return o->st_max; }
const uint32_t* or_outbound_ips(const oracle_t* o, size_t* n) { if (n) *n = o->n_obips; return o->obips; }
int64_t or_window_tmin(const oracle_t* o) { return o->ctmin; }
int64_t or_window_tmax(const oracle_t* o) { return o->ctmax; }
uint64_t or_window_events(const oracle_t* o) { return o->cevents; }
