/*
 * lean_baseline.c — TEST / MEASUREMENT INFRASTRUCTURE, not product code (only bench.py's cpu_baseline leg and tests/ use it).
 *
 * What a careful CPU implementation of the SAME work as K1 would do (BASELINE.md §2 "lean variant"): the join of
 * setFromToV2 (aggregator/data.go:827-870: source must be a pod, destination service first, then pod, then Host label,
 * then raw IP; optional ReverseDirection, datastore/dto.go:226-231) on u32-keyed open-addressing tables instead of the
 * reference's dotted-quad strings and Go maps, followed by per-edge integer aggregation (count, errors, sum, max,
 * sum of squares) in an open-addressing edge table instead of one heap DTO per request (datastore/backend.go:819-847).
 * Single-threaded; one instance per thread/process for a multi-core figure.  Alive records are skipped.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/servicegraph.h"

typedef struct { uint32_t ip, val; } ipent;             /* val = kind << 30 | id, 0 = empty */
typedef struct { uint64_t key, cnt_err, sum, max, ssq; } edgeent;
typedef struct lean {
    ipent* ips; uint32_t ipmask;
    edgeent* edges; uint64_t emask, n_edges;
    uint32_t* pod_of_both; /* unused: an IP in both maps keeps the service id, the pod id in a second table */
    ipent* both; uint32_t bothmask;
    uint64_t accepted, dropped_src, dropped_cap;
    uint32_t max_labels;
} lean;

static uint32_t fmix(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
static uint64_t p2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

lean* lean_create(uint32_t max_ips, uint64_t max_edges, uint32_t max_labels) {
    lean* l = (lean*)calloc(1, sizeof(lean));
    l->ipmask = (uint32_t)p2((uint64_t)max_ips * 2 + 16) - 1; l->ips = (ipent*)calloc((size_t)l->ipmask + 1, sizeof(ipent));
    l->bothmask = 2047; l->both = (ipent*)calloc(2048, sizeof(ipent));
    l->emask = p2(max_edges * 2 + 16) - 1; l->edges = (edgeent*)malloc((size_t)(l->emask + 1) * sizeof(edgeent));
    for (uint64_t i = 0; i <= l->emask; i++) l->edges[i].key = ~0ull;
    l->max_labels = max_labels;
    return l;
}
void lean_destroy(lean* l) { if (!l) return; free(l->ips); free(l->both); free(l->edges); free(l); }

static ipent* ip_slot(ipent* t, uint32_t mask, uint32_t ip) {
    uint32_t h = fmix(ip) & mask;
    while (t[h].val && t[h].ip != ip) h = (h + 1) & mask;
    return &t[h];
}
/* kind 1 = pod, 2 = service (persist.go:55-71, 114-130: ADD/UPDATE upsert) */
void lean_upsert(lean* l, uint32_t ip, uint32_t kind, uint32_t id) {
    ipent* s = ip_slot(l->ips, l->ipmask, ip);
    if (s->val && (s->val >> 30) != kind) {            /* the IP is in both maps: service wins as destination, pod id kept aside */
        const uint32_t svc = kind == 2 ? id : (s->val & 0x3FFFFFFFu), pod = kind == 1 ? id : (s->val & 0x3FFFFFFFu);
        s->val = (3u << 30) | svc;
        ipent* b = ip_slot(l->both, l->bothmask, ip); b->ip = ip; b->val = (1u << 30) | pod;
        return;
    }
    s->ip = ip; s->val = (kind << 30) | id;
}
void lean_reset_window(lean* l) {
    for (uint64_t i = 0; i <= l->emask; i++) l->edges[i].key = ~0ull;
    l->n_edges = 0; l->accepted = l->dropped_src = l->dropped_cap = 0;
}
static int is_err(uint32_t proto, uint32_t status) {
    if (proto == SG_PROTO_HTTP || proto == SG_PROTO_HTTP2) return status >= 500;
    if (proto == SG_PROTO_POSTGRES || proto == SG_PROTO_REDIS || proto == SG_PROTO_MYSQL) return status == 2;
    return 0;
}
size_t lean_process(lean* l, const sg_event* ev, size_t n) {
    size_t acc = 0;
    for (size_t i = 0; i < n; i++) {
        const sg_event* e = &ev[i];
        if (e->flags & SG_EV_ALIVE) continue;
        const ipent* s = ip_slot(l->ips, l->ipmask, e->saddr);
        uint32_t from;
        if (!s->val) { l->dropped_src++; continue; }
        if ((s->val >> 30) == 1) from = s->val & 0x3FFFFFFFu;
        else if ((s->val >> 30) == 3) from = ip_slot(l->both, l->bothmask, e->saddr)->val & 0x3FFFFFFFu;
        else { l->dropped_src++; continue; }                                  /* data.go:829-832 */
        const ipent* d = ip_slot(l->ips, l->ipmask, e->daddr);
        uint32_t to;
        if (d->val) to = d->val & 0x3FFFFFFFu;                                /* service first, else pod (:840-849) */
        else if (e->host_label) { if (e->host_label > l->max_labels) { l->dropped_cap++; continue; } to = SG_MAKE_REF(SG_REF_LABEL, e->host_label - 1); }
        else to = SG_MAKE_REF(SG_REF_OBIP, 0) | (e->daddr & 0x3FFFFFFFu);     /* raw IP (:862-863); good enough for a baseline */
        if (e->flags & SG_EV_REVERSE) { const uint32_t t = from; from = to; to = t; }
        const uint64_t key = ((uint64_t)from << 32) | to;
        uint64_t h = (fmix((uint32_t)key) ^ (fmix(from) * 0x9E3779B1u)) & l->emask;
        while (l->edges[h].key != ~0ull && l->edges[h].key != key) h = (h + 1) & l->emask;
        edgeent* x = &l->edges[h];
        if (x->key == ~0ull) { x->key = key; x->cnt_err = x->sum = x->max = x->ssq = 0; l->n_edges++; }
        const uint64_t dur = e->duration_ns, us = dur / 1000u;
        x->cnt_err += 1ull | ((uint64_t)is_err(e->protocol, e->status) << 32);
        x->sum += dur; if (dur > x->max) x->max = dur; x->ssq += us * us;
        acc++;
    }
    l->accepted += acc;
    return acc;
}
uint64_t lean_edges(const lean* l) { return l->n_edges; }
uint64_t lean_accepted(const lean* l) { return l->accepted; }
uint64_t lean_dropped_src(const lean* l) { return l->dropped_src; }
/* checksums for the parity check against the oracle: sum over edges of count / sum_ns */
void lean_checksums(const lean* l, uint64_t* count, uint64_t* sum_ns, uint64_t* err) {
    uint64_t c = 0, s = 0, er = 0;
    for (uint64_t i = 0; i <= l->emask; i++) if (l->edges[i].key != ~0ull) { c += l->edges[i].cnt_err & 0xFFFFFFFFull; er += l->edges[i].cnt_err >> 32; s += l->edges[i].sum; }
    *count = c; *sum_ns = s; *err = er;
}
