/*
 * http2.c — CPU restatement of the reference's HTTP/2 request assembly (first half).
 * TEST INFRASTRUCTURE ONLY (see sg_oracle.h).
 *
 * Follows aggregator/data.go:
 *   processHttp2Frames   :544-810   frame-header walk (:618-627), first HEADERS frame of an event only (:741,:800),
 *                                   FrameArrival pairing by "pid-fd-streamId" (:545-547), persistReq (:576-616),
 *                                   the one-minute sweep of half-arrived streams (:553-567)
 *   processHttp2Event    :1019-1033 events of pids that are not live are dropped
 *   processExit          :362-377   parsers are dropped by STRING PREFIX of the pid ("12" also drops "123-4")
 *   processTcpConnect    :484-494   a closed connection drops its parser
 *
 * HPACK is golang.org/x/net v0.20.0 (go.mod:118), package http2/hpack — a dependency that is NOT under
 * /root/reference (not vendored).  Restated from RFC 7541 and that package's published behaviour:
 *   Decoder.Write    a block that ends inside a field is saved and prefixed to the next Write; any other
 *                    error abandons the rest of that Write (the reference ignores Write's result)
 *   firstField       set by NewDecoder/Close only; the reference never calls Close between blocks, so a
 *                    dynamic-table size update after the connection's first field is an error once the
 *                    table is non-empty
 *   at()             1..61 static table, then the dynamic table newest-first
 *   huffmanDecode    padding longer than 7 bits, padding that is not all ones, and EOS are errors
 * The payload handed to the decoder is the raw frame payload [offset:endOfFrame): PADDED / PRIORITY
 * fields are NOT stripped (reference behaviour, data.go:731).
 *
 * Pinned by the RFC 7541 Appendix C known-answer vectors (tests/test_http2.py).  The reference holds no
 * test for this path: parity unpinned by reference tests.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sg_oracle.h"

/* ---------------------------------------------------------------- static table (RFC 7541 Appendix A) */
static const char* const ST_NAME[61] = {
    ":authority", ":method", ":method", ":path", ":path", ":scheme", ":scheme", ":status", ":status", ":status",
    ":status", ":status", ":status", ":status", "accept-charset", "accept-encoding", "accept-language",
    "accept-ranges", "accept", "access-control-allow-origin", "age", "allow", "authorization", "cache-control",
    "content-disposition", "content-encoding", "content-language", "content-length", "content-location",
    "content-range", "content-type", "cookie", "date", "etag", "expect", "expires", "from", "host", "if-match",
    "if-modified-since", "if-none-match", "if-range", "if-unmodified-since", "last-modified", "link", "location",
    "max-forwards", "proxy-authenticate", "proxy-authorization", "range", "referer", "refresh", "retry-after",
    "server", "set-cookie", "strict-transport-security", "transfer-encoding", "user-agent", "vary", "via",
    "www-authenticate"};
static const char* const ST_VALUE[61] = {
    "", "GET", "POST", "/", "/index.html", "http", "https", "200", "204", "206", "304", "400", "404", "500", "",
    "gzip, deflate", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "",
    "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", "", ""};

/* ---------------------------------------------------------------- Huffman code (RFC 7541 Appendix B)
 * The code is canonical: within one length the codes are consecutive in symbol order and every shorter
 * code precedes every longer one.  Only the code LENGTH per symbol is tabulated; the codes follow.
 * (Kraft sum of the table below is exactly 1 with EOS = 30 ones; checked by or_hpack_selfcheck.) */
static uint8_t HUFF_LEN[257];
static uint16_t HUFF_SORTED[257];         /* symbols ordered by (length, symbol) */
static uint16_t HUFF_COUNT[31];           /* symbols per length */
static int huff_ready = 0;

static void huff_set(uint8_t len, const int* syms, size_t n) { for (size_t i = 0; i < n; i++) HUFF_LEN[syms[i]] = len; }
#define HSET(len, ...) do { static const int s_[] = {__VA_ARGS__}; huff_set(len, s_, sizeof s_ / sizeof s_[0]); } while (0)

static void huff_init(void) {
    if (huff_ready) return;
    memset(HUFF_LEN, 0, sizeof HUFF_LEN);
    HSET(5, '0', '1', '2', 'a', 'c', 'e', 'i', 'o', 's', 't');
    HSET(6, ' ', '%', '-', '.', '/', '3', '4', '5', '6', '7', '8', '9', '=', 'A', '_', 'b', 'd', 'f', 'g', 'h', 'l', 'm', 'n', 'p', 'r', 'u');
    HSET(7, ':', 'B', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'J', 'K', 'L', 'M', 'N', 'O', 'P', 'Q', 'R', 'S', 'T', 'U', 'V', 'W', 'Y',
         'j', 'k', 'q', 'v', 'w', 'x', 'y', 'z');
    HSET(8, '&', '*', ',', ';', 'X', 'Z');
    HSET(10, '!', '"', '(', ')', '?');
    HSET(11, '\'', '+', '|');
    HSET(12, '#', '>');
    HSET(13, 0, '$', '@', '[', ']', '~');
    HSET(14, '^', '}');
    HSET(15, '<', '`', '{');
    HSET(19, '\\', 195, 208);
    HSET(20, 128, 130, 131, 162, 184, 194, 224, 226);
    HSET(21, 153, 161, 167, 172, 176, 177, 179, 209, 216, 217, 227, 229, 230);
    HSET(22, 129, 132, 133, 134, 136, 146, 154, 156, 160, 163, 164, 169, 170, 173, 178, 181, 185, 186, 187, 189, 190, 196, 198, 228, 232, 233);
    HSET(23, 1, 135, 137, 138, 139, 140, 141, 143, 147, 149, 150, 151, 152, 155, 157, 158, 165, 166, 168, 174, 175, 180, 182, 183, 188,
         191, 197, 231, 239);
    HSET(24, 9, 142, 144, 145, 148, 159, 171, 206, 215, 225, 236, 237);
    HSET(25, 199, 207, 234, 235);
    HSET(26, 192, 193, 200, 201, 202, 205, 210, 213, 218, 219, 238, 240, 242, 243, 255);
    HSET(27, 203, 204, 211, 212, 214, 221, 222, 223, 241, 244, 245, 246, 247, 248, 250, 251, 252, 253, 254);
    HSET(28, 2, 3, 4, 5, 6, 7, 8, 11, 12, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24, 25, 26, 27, 28, 29, 30, 31, 127, 220, 249);
    HSET(30, 10, 13, 22, 256);
    memset(HUFF_COUNT, 0, sizeof HUFF_COUNT);
    size_t k = 0;
    for (int len = 1; len <= 30; len++)
        for (int s = 0; s < 257; s++)
            if (HUFF_LEN[s] == len) { HUFF_SORTED[k++] = (uint16_t)s; HUFF_COUNT[len]++; }
    huff_ready = 1;
}

/* 0 when every symbol has a length and the Kraft sum is exactly 1 */
int or_hpack_selfcheck(void) {
    huff_init();
    uint64_t kraft = 0;
    for (int s = 0; s < 257; s++) { if (HUFF_LEN[s] == 0) return -1; kraft += 1ull << (30 - HUFF_LEN[s]); }
    return kraft == (1ull << 30) ? 0 : -2;
}

/* code of a symbol (for the encoder used by the tests' payload builders) */
uint32_t or_hpack_huff_code(int sym, uint8_t* len_out) {
    huff_init();
    uint32_t code = 0, first = 0; size_t idx = 0;
    for (int len = 1; len <= 30; len++) {
        for (uint16_t j = 0; j < HUFF_COUNT[len]; j++)
            if (HUFF_SORTED[idx + j] == sym) { *len_out = (uint8_t)len; return first + j; }
        idx += HUFF_COUNT[len]; first = (first + HUFF_COUNT[len]) << 1;
    }
    (void)code; *len_out = 0; return 0;
}

/* hpack.huffmanDecode (maxLen 0).  Returns the decoded length, or -1 (ErrInvalidHuffman). out has room for 2n bytes
 * (shortest code is 5 bits). */
long or_hpack_huff_decode(const uint8_t* v, size_t n, uint8_t* out) {
    huff_init();
    size_t o = 0;
    uint32_t code = 0, first = 0; size_t idx = 0; int len = 0; int all_ones = 1;
    for (size_t i = 0; i < n; i++) {
        for (int b = 7; b >= 0; b--) {
            const uint32_t bit = (v[i] >> b) & 1u;
            code = (code << 1) | bit; len++; if (!bit) all_ones = 0;
            const uint32_t cnt = HUFF_COUNT[len];
            if (code - first < cnt) {
                const uint16_t sym = HUFF_SORTED[idx + (code - first)];
                if (sym == 256) return -1;                         /* EOS inside the string */
                out[o++] = (uint8_t)sym;
                code = 0; first = 0; idx = 0; len = 0; all_ones = 1;
            } else {
                idx += cnt; first = (first + cnt) << 1;
                if (len == 30) return -1;                            /* unreachable for a complete code */
            }
        }
    }
    if (len > 7) return -1;                                          /* incomplete symbol / over-long padding */
    if (!all_ones) return -1;                                        /* padding must be a prefix of EOS */
    return (long)o;
}

/* ---------------------------------------------------------------- decoder */
typedef struct { uint8_t* name; uint8_t* value; size_t nlen, vlen; } hp_field;

struct or_hpack {
    hp_field* ents; size_t n, cap;            /* dynamic table, oldest first */
    uint32_t size, max_size, allowed_max;
    uint8_t* save; size_t save_n, save_cap;
    int first_field;
    or_hpack_emit_fn emit; void* ctx;
};

enum { HP_OK = 0, HP_NEED_MORE = 1, HP_ERR = -1 };

or_hpack* or_hpack_create(uint32_t max_table_size) {
    or_hpack* d = calloc(1, sizeof *d);
    d->max_size = d->allowed_max = max_table_size; d->first_field = 1;
    return d;
}
static void field_free(hp_field* f) { free(f->name); free(f->value); }
void or_hpack_destroy(or_hpack* d) {
    if (!d) return;
    for (size_t i = 0; i < d->n; i++) field_free(&d->ents[i]);
    free(d->ents); free(d->save); free(d);
}
void or_hpack_set_emit(or_hpack* d, or_hpack_emit_fn fn, void* ctx) { d->emit = fn; d->ctx = ctx; }
size_t or_hpack_dyn_len(const or_hpack* d) { return d->n; }
uint32_t or_hpack_dyn_size(const or_hpack* d) { return d->size; }
/* i = 0 is the newest entry */
int or_hpack_dyn_at(const or_hpack* d, size_t i, const uint8_t** name, size_t* nlen, const uint8_t** value, size_t* vlen) {
    if (i >= d->n) return -1;
    const hp_field* f = &d->ents[d->n - 1 - i];
    *name = f->name; *nlen = f->nlen; *value = f->value; *vlen = f->vlen;
    return 0;
}

static uint8_t* dup_bytes(const uint8_t* p, size_t n) { uint8_t* r = malloc(n ? n : 1); if (n) memcpy(r, p, n); return r; }

static void dyn_evict(or_hpack* d) {                                 /* dynamicTable.evict */
    size_t k = 0;
    while (d->size > d->max_size && k < d->n) { d->size -= (uint32_t)(d->ents[k].nlen + d->ents[k].vlen + 32); k++; }
    for (size_t i = 0; i < k; i++) field_free(&d->ents[i]);
    if (k) { memmove(d->ents, d->ents + k, (d->n - k) * sizeof(hp_field)); d->n -= k; }
}
static void dyn_add(or_hpack* d, const uint8_t* name, size_t nlen, const uint8_t* value, size_t vlen) {
    if (d->n == d->cap) { d->cap = d->cap ? d->cap * 2 : 16; d->ents = realloc(d->ents, d->cap * sizeof(hp_field)); }
    hp_field* f = &d->ents[d->n++];
    f->name = dup_bytes(name, nlen); f->nlen = nlen; f->value = dup_bytes(value, vlen); f->vlen = vlen;
    d->size += (uint32_t)(nlen + vlen + 32);
    dyn_evict(d);
}

/* Decoder.at */
static int table_at(const or_hpack* d, uint64_t i, const uint8_t** name, size_t* nlen, const uint8_t** value, size_t* vlen) {
    if (i == 0) return -1;
    if (i <= 61) {
        *name = (const uint8_t*)ST_NAME[i - 1]; *nlen = strlen(ST_NAME[i - 1]);
        *value = (const uint8_t*)ST_VALUE[i - 1]; *vlen = strlen(ST_VALUE[i - 1]);
        return 0;
    }
    if (i > (uint64_t)d->n + 61) return -1;
    const hp_field* f = &d->ents[d->n - (size_t)(i - 61)];
    *name = f->name; *nlen = f->nlen; *value = f->value; *vlen = f->vlen;
    return 0;
}

/* hpack.readVarInt: consumed bytes in *used */
static int read_varint(unsigned nbits, const uint8_t* p, size_t len, uint64_t* out, size_t* used) {
    if (len == 0) return HP_NEED_MORE;
    uint64_t i = p[0];
    if (nbits < 8) i &= (1ull << nbits) - 1;
    if (i < (1ull << nbits) - 1) { *out = i; *used = 1; return HP_OK; }
    size_t k = 1; uint64_t m = 0;
    while (k < len) {
        const uint8_t b = p[k++];
        i += (uint64_t)(b & 127) << m;
        if ((b & 128) == 0) { *out = i; *used = k; return HP_OK; }
        m += 7;
        if (m >= 63) return HP_ERR;                                  /* errVarintOverflow */
    }
    return HP_NEED_MORE;
}

typedef struct { const uint8_t* b; size_t n; int huff; } undecoded;

/* Decoder.readString (maxStrLen == 0: no limit) */
static int read_string(const uint8_t* p, size_t len, undecoded* u, size_t* used) {
    if (len == 0) return HP_NEED_MORE;
    const int huff = (p[0] & 128) != 0;
    uint64_t slen; size_t k;
    int e = read_varint(7, p, len, &slen, &k);
    if (e != HP_OK) return e;
    if ((uint64_t)(len - k) < slen) return HP_NEED_MORE;
    u->b = p + k; u->n = (size_t)slen; u->huff = huff; *used = k + (size_t)slen;
    return HP_OK;
}

/* Decoder.decodeString: result malloc'd */
static int decode_string(const undecoded* u, uint8_t** out, size_t* outlen) {
    if (!u->huff) { *out = dup_bytes(u->b, u->n); *outlen = u->n; return HP_OK; }
    uint8_t* buf = malloc(u->n * 2 + 1);
    long r = or_hpack_huff_decode(u->b, u->n, buf);
    if (r < 0) { free(buf); return HP_ERR; }
    *out = buf; *outlen = (size_t)r; return HP_OK;
}

static void call_emit(or_hpack* d, const uint8_t* name, size_t nlen, const uint8_t* value, size_t vlen) {
    if (d->emit) d->emit(d->ctx, name, nlen, value, vlen);
}

/* Decoder.parseHeaderFieldRepr: one field at p; *used = bytes consumed on HP_OK */
static int parse_field(or_hpack* d, const uint8_t* p, size_t len, size_t* used) {
    const uint8_t b = p[0];
    if (b & 128) {                                                   /* 6.1 indexed */
        uint64_t idx; size_t k;
        int e = read_varint(7, p, len, &idx, &k); if (e != HP_OK) return e;
        const uint8_t *nm, *val; size_t nl, vl;
        if (table_at(d, idx, &nm, &nl, &val, &vl) != 0) return HP_ERR;
        *used = k;
        /* the emit callback may not outlive a table change: copy */
        uint8_t* n2 = dup_bytes(nm, nl); uint8_t* v2 = dup_bytes(val, vl);
        call_emit(d, n2, nl, v2, vl); free(n2); free(v2);
        return HP_OK;
    }
    unsigned nbits; int indexed;
    if ((b & 192) == 64) { nbits = 6; indexed = 1; }                 /* 6.2.1 incremental indexing */
    else if ((b & 240) == 0 || (b & 240) == 16) { nbits = 4; indexed = 0; }   /* 6.2.2 / 6.2.3 */
    else if ((b & 224) == 32) {                                      /* 6.3 dynamic table size update */
        if (!d->first_field && d->size > 0) return HP_ERR;
        uint64_t size; size_t k;
        int e = read_varint(5, p, len, &size, &k); if (e != HP_OK) return e;
        if (size > d->allowed_max) return HP_ERR;
        d->max_size = (uint32_t)size; dyn_evict(d);
        *used = k; return HP_OK;
    } else return HP_ERR;

    uint64_t name_idx; size_t k;
    int e = read_varint(nbits, p, len, &name_idx, &k); if (e != HP_OK) return e;
    const uint8_t* nm = NULL; size_t nl = 0; uint8_t* own_name = NULL;
    undecoded uname = {0, 0, 0}, uval;
    if (name_idx > 0) {
        const uint8_t* val; size_t vl;
        if (table_at(d, name_idx, &nm, &nl, &val, &vl) != 0) return HP_ERR;
    } else {
        size_t u; e = read_string(p + k, len - k, &uname, &u); if (e != HP_OK) return e;
        k += u;
    }
    size_t u; e = read_string(p + k, len - k, &uval, &u); if (e != HP_OK) return e;
    k += u;
    if (name_idx == 0) {
        if (decode_string(&uname, &own_name, &nl) != HP_OK) return HP_ERR;
    } else own_name = dup_bytes(nm, nl);                             /* table storage may move in dyn_add */
    uint8_t* value; size_t vl;
    if (decode_string(&uval, &value, &vl) != HP_OK) { free(own_name); return HP_ERR; }
    *used = k;
    if (indexed) dyn_add(d, own_name, nl, value, vl);
    call_emit(d, own_name, nl, value, vl);
    free(own_name); free(value);
    return HP_OK;
}

/* Decoder.Write */
int or_hpack_write(or_hpack* d, const uint8_t* p, size_t n) {
    if (n == 0) return 0;
    const uint8_t* buf = p; size_t len = n; uint8_t* joined = NULL;
    if (d->save_n) {
        joined = malloc(d->save_n + n);
        memcpy(joined, d->save, d->save_n); memcpy(joined + d->save_n, p, n);
        buf = joined; len = d->save_n + n; d->save_n = 0;
    }
    int err = 0;
    while (len > 0) {
        size_t used = 0;
        err = parse_field(d, buf, len, &used);
        if (err == HP_NEED_MORE) {
            if (d->save_cap < len) { d->save_cap = len * 2; d->save = realloc(d->save, d->save_cap); }
            memmove(d->save, buf, len); d->save_n = len;
            free(joined); return 0;
        }
        d->first_field = 0;
        if (err != HP_OK) break;
        buf += used; len -= used;
    }
    free(joined);
    return err;
}

/* ---------------------------------------------------------------- frame assembly */
typedef struct { char* p; size_t n; } gostr;
static void gs_set(gostr* s, const uint8_t* v, size_t n) { free(s->p); s->p = malloc(n + 1); memcpy(s->p, v, n); s->p[n] = 0; s->n = n; }
static void gs_free(gostr* s) { free(s->p); s->p = NULL; s->n = 0; }

typedef struct frame_arrival {
    uint32_t pid; uint64_t fd; uint32_t stream;
    int client, server;
    gostr method, path, to_uid; int grpc;      /* datastore.Request fields the header callbacks fill */
    uint64_t latency;                           /* req.Latency: the client frame's write time */
    uint32_t status_code, grpc_status;
    struct frame_arrival* next;
} frame_arrival;

typedef struct h2_parser {
    uint32_t pid; uint64_t fd;
    or_hpack *client, *server;
    struct h2_parser* next;
} h2_parser;

#define H2_BUCKETS 4096
struct or_h2 {
    frame_arrival* frames[H2_BUCKETS]; size_t n_frames;
    h2_parser* parsers[H2_BUCKETS]; size_t n_parsers;
    uint32_t* live; size_t n_live, cap_live;
    uint64_t dropped_not_live, dropped_unparsed, dropped_time;
};

or_h2* or_h2_create(void) { return calloc(1, sizeof(or_h2)); }
static void fa_free(frame_arrival* f) { gs_free(&f->method); gs_free(&f->path); gs_free(&f->to_uid); free(f); }
static void parser_free(h2_parser* p) { or_hpack_destroy(p->client); or_hpack_destroy(p->server); free(p); }
void or_h2_destroy(or_h2* h) {
    if (!h) return;
    for (size_t b = 0; b < H2_BUCKETS; b++) {
        for (frame_arrival* f = h->frames[b]; f;) { frame_arrival* nx = f->next; fa_free(f); f = nx; }
        for (h2_parser* p = h->parsers[b]; p;) { h2_parser* nx = p->next; parser_free(p); p = nx; }
    }
    free(h->live); free(h);
}
size_t or_h2_pending(const or_h2* h) { return h->n_frames; }
size_t or_h2_parsers(const or_h2* h) { return h->n_parsers; }
uint64_t or_h2_dropped_not_live(const or_h2* h) { return h->dropped_not_live; }
uint64_t or_h2_dropped_unparsed(const or_h2* h) { return h->dropped_unparsed; }

static size_t h2_hash(uint32_t pid, uint64_t fd, uint32_t stream) {
    uint64_t x = ((uint64_t)pid << 32) ^ fd * 0x9E3779B97F4A7C15ull ^ (uint64_t)stream * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return (size_t)(x % H2_BUCKETS);
}

/* liveProcesses (data.go:86): processExec :354-360 adds, the periodic /proc scan :199-217 and exit remove */
void or_h2_proc_exec(or_h2* h, uint32_t pid) {
    for (size_t i = 0; i < h->n_live; i++) if (h->live[i] == pid) return;
    if (h->n_live == h->cap_live) { h->cap_live = h->cap_live ? h->cap_live * 2 : 16; h->live = realloc(h->live, h->cap_live * sizeof(uint32_t)); }
    h->live[h->n_live++] = pid;
}
static int is_live(const or_h2* h, uint32_t pid) { for (size_t i = 0; i < h->n_live; i++) if (h->live[i] == pid) return 1; return 0; }

/* processExit :362-377 (clearProc removes the pid from liveProcesses via the same path, cluster.go) */
void or_h2_proc_exit(or_h2* h, uint32_t pid) {
    for (size_t i = 0; i < h->n_live; i++) if (h->live[i] == pid) { h->live[i] = h->live[--h->n_live]; break; }
    char pid_s[16], key[48]; snprintf(pid_s, sizeof pid_s, "%u", pid);
    const size_t pl = strlen(pid_s);
    for (size_t b = 0; b < H2_BUCKETS; b++) {
        h2_parser** pp = &h->parsers[b];
        while (*pp) {
            h2_parser* p = *pp;
            snprintf(key, sizeof key, "%u-%llu", p->pid, (unsigned long long)p->fd);
            if (strncmp(key, pid_s, pl) == 0) { *pp = p->next; parser_free(p); h->n_parsers--; }   /* strings.HasPrefix(key, pid_s) */
            else pp = &p->next;
        }
    }
}

/* processTcpConnect, closed branch :484-494 */
void or_h2_conn_closed(or_h2* h, uint32_t pid, uint64_t fd) {
    h2_parser** pp = &h->parsers[h2_hash(pid, fd, 0)];
    while (*pp) {
        h2_parser* p = *pp;
        if (p->pid == pid && p->fd == fd) { *pp = p->next; parser_free(p); h->n_parsers--; return; }
        pp = &p->next;
    }
}

/* the one-minute ticker of processHttp2Frames :553-567 */
void or_h2_sweep(or_h2* h) {
    for (size_t b = 0; b < H2_BUCKETS; b++) {
        frame_arrival** pp = &h->frames[b];
        while (*pp) {
            frame_arrival* f = *pp;
            if (f->client != f->server) { *pp = f->next; fa_free(f); h->n_frames--; }
            else pp = &f->next;
        }
    }
}

static h2_parser* parser_of(or_h2* h, uint32_t pid, uint64_t fd) {
    const size_t b = h2_hash(pid, fd, 0);
    for (h2_parser* p = h->parsers[b]; p; p = p->next) if (p->pid == pid && p->fd == fd) return p;
    h2_parser* p = calloc(1, sizeof *p);
    p->pid = pid; p->fd = fd; p->client = or_hpack_create(4096); p->server = or_hpack_create(4096);
    p->next = h->parsers[b]; h->parsers[b] = p; h->n_parsers++;
    return p;
}
static frame_arrival* frame_of(or_h2* h, uint32_t pid, uint64_t fd, uint32_t stream) {
    const size_t b = h2_hash(pid, fd, stream + 1);
    for (frame_arrival* f = h->frames[b]; f; f = f->next) if (f->pid == pid && f->fd == fd && f->stream == stream) return f;
    frame_arrival* f = calloc(1, sizeof *f);
    f->pid = pid; f->fd = fd; f->stream = stream;
    f->next = h->frames[b]; h->frames[b] = f; h->n_frames++;
    return f;
}
static void frame_delete(or_h2* h, frame_arrival* f) {
    frame_arrival** pp = &h->frames[h2_hash(f->pid, f->fd, f->stream + 1)];
    while (*pp && *pp != f) pp = &(*pp)->next;
    if (*pp) { *pp = f->next; fa_free(f); h->n_frames--; }
}

static int name_is(const uint8_t* n, size_t nl, const char* lit) { return nl == strlen(lit) && memcmp(n, lit, nl) == 0; }

/* strconv.Atoi with the error dropped (`s, _ := strconv.Atoi(v)`), then uint32(s) */
uint32_t or_go_atoi_u32(const uint8_t* v, size_t n) {
    if (n == 0) return 0;
    size_t i = 0; int neg = 0;
    if (v[0] == '+' || v[0] == '-') { neg = v[0] == '-'; i = 1; if (n == 1) return 0; }
    uint64_t acc = 0; int range = 0;
    for (; i < n; i++) {
        if (v[i] < '0' || v[i] > '9') return 0;                      /* ErrSyntax => 0 */
        if (!range) {
            if (acc > (UINT64_MAX - 9) / 10) range = 1; else acc = acc * 10 + (uint64_t)(v[i] - '0');
        }
    }
    const uint64_t lim = neg ? (1ull << 63) : (1ull << 63) - 1;
    if (range || acc > lim) acc = lim;                               /* ErrRange => clamped value is returned */
    const int64_t s = neg ? (int64_t)(0 - acc) : (int64_t)acc;
    return (uint32_t)(uint64_t)s;
}

static void on_client_header(void* ctx, const uint8_t* n, size_t nl, const uint8_t* v, size_t vl) {   /* reqHeaderSet :707-730 */
    frame_arrival* f = ctx;
    if (name_is(n, nl, ":method")) { if (f->method.n == 0) gs_set(&f->method, v, vl); }
    else if (name_is(n, nl, ":path")) { if (f->path.n == 0) gs_set(&f->path, v, vl); }
    else if (name_is(n, nl, ":authority")) { if (f->to_uid.n == 0) gs_set(&f->to_uid, v, vl); }
    else if (name_is(n, nl, "content-type")) { if (!f->grpc && vl >= 16 && memcmp(v, "application/grpc", 16) == 0) f->grpc = 1; }
}
static void on_server_header(void* ctx, const uint8_t* n, size_t nl, const uint8_t* v, size_t vl) {   /* respHeaderSet :780-792 */
    frame_arrival* f = ctx;
    if (name_is(n, nl, ":status")) f->status_code = or_go_atoi_u32(v, vl);
    else if (name_is(n, nl, "grpc-status")) f->grpc_status = or_go_atoi_u32(v, vl);
}

static void copy_cap(char* dst, size_t cap, const gostr* s) {
    size_t n = s->n < cap - 1 ? s->n : cap - 1;
    if (n) memcpy(dst, s->p, n);
    dst[n] = 0;
}

/* persistReq :576-616, up to (not including) the join.  Returns 1 when the request goes on. */
static int persist_req(or_h2* h, const frame_arrival* f, uint64_t write_ns, int tls, or_h2_out* out) {
    if (f->method.n == 0 || f->path.n == 0) { h->dropped_unparsed++; return 0; }
    const uint64_t lat = write_ns - f->latency;                      /* uint64 wrap as in Go */
    if (write_ns < lat) { h->dropped_time++; return 0; }             /* :608-611 (compares against the difference) */
    memset(out, 0, sizeof *out);
    copy_cap(out->method, sizeof out->method, &f->method);
    copy_cap(out->path, sizeof out->path, &f->path);
    copy_cap(out->authority, sizeof out->authority, &f->to_uid);
    if (f->grpc) { strcpy(out->protocol, "gRPC"); out->status_code = f->grpc_status; }
    else { strcpy(out->protocol, tls ? "HTTPS" : "HTTP2"); out->status_code = f->status_code; }
    out->latency = lat;
    return 1;
}

/* one HTTP2 L7 event (method_id 1 = CLIENT_FRAME, 2 = SERVER_FRAME).  Returns 1 and fills *out when a
 * request is complete and passes persistReq's own checks. */
int or_h2_event(or_h2* h, uint32_t pid, uint64_t fd, int method_id, const uint8_t* payload, uint32_t size,
                uint64_t write_ns, int tls, or_h2_out* out) {
    if (!is_live(h, pid)) { h->dropped_not_live++; return 0; }       /* processHttp2Event :1023-1029 */
    h2_parser* ps = parser_of(h, pid, fd);                           /* created before the direction test, :646-657 */
    if (method_id != 1 && method_id != 2) return 0;                  /* "unknown http2 frame type" :802-805 */
    const uint8_t* buf = payload; const long len = (long)size;
    uint32_t offset = 0;
    for (;;) {
        if (len - (long)offset < 9) break;
        const uint8_t* fh = buf + offset;
        const uint32_t flen = (uint32_t)fh[0] << 16 | (uint32_t)fh[1] << 8 | fh[2];
        const uint8_t ftype = fh[3];
        const uint32_t stream = ((uint32_t)fh[5] << 24 | (uint32_t)fh[6] << 16 | (uint32_t)fh[7] << 8 | fh[8]) & 0x7FFFFFFFu;
        offset += 9;
        const uint32_t end = offset + flen;
        if (len < (long)end) break;
        if (ftype != 1) { offset = end; continue; }                  /* http2.FrameHeaders == 0x1 */
        frame_arrival* f = frame_of(h, pid, fd, stream);
        if (method_id == 1) {
            f->client = 1; f->latency = write_ns;
            or_hpack_set_emit(ps->client, on_client_header, f);
            (void)or_hpack_write(ps->client, buf + offset, flen);
            if (f->server) { int r = persist_req(h, f, write_ns, tls, out); frame_delete(h, f); return r; }
        } else {
            f->server = 1;
            or_hpack_set_emit(ps->server, on_server_header, f);
            (void)or_hpack_write(ps->server, buf + offset, flen);
            if (f->client) { int r = persist_req(h, f, write_ns, tls, out); frame_delete(h, f); return r; }
        }
        break;                                                       /* only the first HEADERS frame of an event */
    }
    return 0;
}
