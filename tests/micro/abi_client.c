/* abi_client.c — a plain C11 client of include/servicegraph.h, the way a cgo / FFI binding sees the library: create an
 * engine, announce two pods and a service (sg_upsert_*), ingest a handful of packed events (one of them from an unknown
 * source, one to an outbound Host label), close the window (sg_flush_window) and check the rows.
 *   gcc -std=c11 -Wall -Werror -Iinclude tests/micro/abi_client.c -Lalaz_amd/lib -lservicegraph -o abi_client
 * Exit code 0 = every check passed, 77 = no usable MI355X (sg_create returned SG_ENODEV: there is no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "servicegraph.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "abi_client: check failed at line %d: %s (%s)\n", __LINE__, #c, h ? sg_last_error(h) : ""); return 1; } } while (0)

static uint32_t ip(unsigned a, unsigned b, unsigned c, unsigned d) { return (a << 24) | (b << 16) | (c << 8) | d; }

int main(void) {
    sg_handle h = NULL;
    CHECK(sg_abi_version() == SG_ABI_VERSION);
    sg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (uint32_t)sizeof cfg; cfg.abi_version = SG_ABI_VERSION;
    cfg.max_known_nodes = 64; cfg.max_labels = 16; cfg.max_outbound_ips = 16; cfg.max_ips = 64; cfg.max_edges = 1024;
    cfg.max_batch = 4096; cfg.layers = 1; cfg.world = 1; cfg.windows_in_flight = 1;
    int rc = sg_create(&cfg, &h);
    if (rc == SG_ENODEV) { fprintf(stderr, "abi_client: no usable gfx950 device\n"); return 77; }
    CHECK(rc == SG_OK && h != NULL);

    const size_t nw = sg_weights_count(1);
    float* w = (float*)malloc(nw * sizeof *w);
    CHECK(w != NULL);
    for (size_t i = 0; i < nw; i++) w[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f * 0.2f - 0.1f;
    CHECK(sg_load_weights(h, w, nw) == SG_OK);
    free(w);
    CHECK(sg_set_clock(h, 1000000000ull, 1700000000000000000ull) == SG_OK);

    CHECK(sg_upsert_pod(h, ip(10, 0, 0, 1), 0) == SG_OK);
    CHECK(sg_upsert_pod(h, ip(10, 0, 0, 2), 1) == SG_OK);
    CHECK(sg_upsert_service(h, ip(172, 16, 0, 1), 2) == SG_OK);
    CHECK(sg_set_label_count(h, 1) == SG_OK);

    sg_event ev[6];
    memset(ev, 0, sizeof ev);
    for (int i = 0; i < 6; i++) { ev[i].protocol = SG_PROTO_HTTP; ev[i].status = 200; ev[i].duration_ns = 1000000u * (uint64_t)(i + 1); ev[i].write_time_ns = 2000000000ull + (uint64_t)i; }
    ev[0].saddr = ip(10, 0, 0, 1); ev[0].daddr = ip(172, 16, 0, 1);                           /* pod 0 -> service 2 */
    ev[1].saddr = ip(10, 0, 0, 1); ev[1].daddr = ip(172, 16, 0, 1); ev[1].status = 503;       /* ... an error */
    ev[2].saddr = ip(10, 0, 0, 2); ev[2].daddr = ip(10, 0, 0, 1);                             /* pod 1 -> pod 0 */
    ev[3].saddr = ip(192, 168, 0, 9); ev[3].daddr = ip(10, 0, 0, 1);                          /* unknown source: dropped (data.go:829-832) */
    ev[4].saddr = ip(10, 0, 0, 2); ev[4].daddr = ip(8, 8, 8, 8); ev[4].host_label = 1;        /* outbound, named by Host header */
    ev[5].saddr = ip(10, 0, 0, 2); ev[5].daddr = ip(9, 9, 9, 9);                              /* outbound, named by its IP */
    CHECK(sg_ingest(h, ev, 6) == SG_OK);

    sg_edge_out rows[16];
    size_t n = 0;
    CHECK(sg_flush_window(h, 0, rows, 16, &n) == SG_OK);
    CHECK(n == 4);
    /* rows are sorted by (dense(from), dense(to)): (0 -> 2), (1 -> 0), (1 -> label 0), (1 -> obip 0) */
    CHECK(rows[0].from_ref == SG_MAKE_REF(SG_REF_KNOWN, 0) && rows[0].to_ref == SG_MAKE_REF(SG_REF_KNOWN, 2));
    CHECK(rows[0].count == 2 && rows[0].err_count == 1 && rows[0].sum_ns == 3000000ull && rows[0].max_ns == 2000000ull && rows[0].sumsq_us == 1000ull * 1000 + 2000ull * 2000);
    CHECK(rows[1].from_ref == SG_MAKE_REF(SG_REF_KNOWN, 1) && rows[1].to_ref == SG_MAKE_REF(SG_REF_KNOWN, 0) && rows[1].count == 1);
    CHECK(rows[2].to_ref == SG_MAKE_REF(SG_REF_LABEL, 0) && rows[2].count == 1);
    CHECK(rows[3].to_ref == SG_MAKE_REF(SG_REF_OBIP, 0) && rows[3].count == 1);
    for (size_t i = 0; i < n; i++) CHECK(rows[i].score > 0.0f && rows[i].score < 1.0f);
    CHECK(rows[0].err_ratio == 0.5f);

    uint32_t ob[4]; size_t nob = 0;
    CHECK(sg_window_outbound_ips(h, ob, 4, &nob) == SG_OK && nob == 1 && ob[0] == ip(9, 9, 9, 9));
    sg_stats st;
    CHECK(sg_stats_get(h, &st) == SG_OK);
    CHECK(st.events_in == 6 && st.events_dropped_src == 1 && st.last_window_events == 5 && st.last_window_edges == 4 && st.windows == 1);
    sg_geometry geo;
    CHECK(sg_geometry_get(h, &geo) == SG_OK && geo.partitions >= 64);

    /* the view form: rows stay in the engine's page-locked buffer */
    CHECK(sg_ingest(h, ev, 3) == SG_OK);
    const sg_edge_out* view = NULL;
    CHECK(sg_flush_window_view(h, 1000, &view, &n) == SG_OK && n == 2 && view != NULL && view[0].count == 2);
    /* the close in two halves: begin marks the boundary, what is ingested afterwards is the next window's */
    CHECK(sg_flush_end_view(h, &view, &n) == SG_ESTATE);
    CHECK(sg_ingest(h, ev, 3) == SG_OK);
    CHECK(sg_flush_begin(h, 2000) == SG_OK);
    CHECK(sg_flush_begin(h, 2000) == SG_ESTATE);
    CHECK(sg_ingest(h, ev + 4, 1) == SG_OK);                                            /* arrives while the window closes: the next one's */
    CHECK(sg_flush_end_view(h, &view, &n) == SG_OK && n == 2 && view[0].count == 2);
    CHECK(sg_flush_begin(h, 3000) == SG_OK);
    CHECK(sg_flush_end(h, rows, 8, &n) == SG_OK && n == 1 && rows[0].count == 1);
    CHECK(sg_destroy(h) == SG_OK);
    h = NULL;

    /* ABI 5 — warm windows by name (an engine of this size would not keep the state by itself): the 8-byte-record path (k1_variant 3) with
     * SG_CFG_WARM.  The first window is rebuilt, the same requests again and a subset of them are closed out of the kept edge set; the rows
     * are the same either way; sg_set_warm(h, 0) sends every window down the rebuild again. */
    cfg.k1_variant = 3; cfg.flags = SG_CFG_WARM;
    CHECK(sg_create(&cfg, &h) == SG_OK && h != NULL);
    w = (float*)malloc(nw * sizeof *w);
    CHECK(w != NULL);
    for (size_t i = 0; i < nw; i++) w[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.0f * 0.2f - 0.1f;
    CHECK(sg_load_weights(h, w, nw) == SG_OK);
    free(w);
    CHECK(sg_geometry_get(h, &geo) == SG_OK && geo.k1_narrow == 1 && geo.warm_windows == 1);
    CHECK(sg_upsert_pod(h, ip(10, 0, 0, 1), 0) == SG_OK && sg_upsert_pod(h, ip(10, 0, 0, 2), 1) == SG_OK && sg_upsert_service(h, ip(172, 16, 0, 1), 2) == SG_OK);
    CHECK(sg_set_label_count(h, 1) == SG_OK);
    sg_edge_out first[4];
    for (int win = 0; win < 4; win++) {
        if (win == 3) CHECK(sg_set_warm(h, 0) == SG_OK);
        const size_t feed = win == 2 ? 2 : 3;                                             /* window 2: only pod 0 -> service 2 */
        CHECK(sg_ingest(h, ev, feed) == SG_OK);
        CHECK(sg_flush_window(h, 0, rows, 16, &n) == SG_OK);
        CHECK(n == (win == 2 ? 1u : 2u) && rows[0].count == 2 && rows[0].err_count == 1 && rows[0].sum_ns == 3000000ull);
        if (win == 0) memcpy(first, rows, 2 * sizeof rows[0]);
        else if (win != 2) CHECK(memcmp(first, rows, 2 * sizeof rows[0]) == 0);           /* warm or rebuilt: the same bytes */
    }
    CHECK(sg_stats_get(h, &st) == SG_OK && st.windows == 4 && st.windows_warm == 2 && st.windows_cold == 2);
    CHECK(sg_destroy(h) == SG_OK);
    printf("abi_client ok\n");
    return 0;
}
