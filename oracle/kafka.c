/*
 * kafka.c — CPU restatement of the reference's Kafka payload decode (second half).
 * TEST INFRASTRUCTURE ONLY (see sg_oracle.h).
 *
 * Follows aggregator/data.go:929-1017 (decodeKafkaPayload: Produce request / Fetch response -> one message per
 * record of every RecordBatch; any error, and any panic — recovered at :943-948 — yields no message at all) and
 * the Sarama-derived package it calls:
 *   aggregator/kafka/request.go:28-62,186-215      size prefix, request header, only api key 0 is known
 *   aggregator/kafka/produce_request.go:29-89      topics -> partitions -> Records (Go maps: a repeated topic
 *                                                   resets its partitions, a repeated partition overwrites)
 *   aggregator/kafka/response_header.go:295-313, fetch_response.go:41-144,163-214
 *   aggregator/kafka/records.go:46-71              magic byte at offset 16 picks MessageSet (<2) or RecordBatch
 *   aggregator/kafka/record_batch.go:51-139        header, CRC-32C, partial trailing batch, decompress, records
 *   aggregator/kafka/record.go:41-87               varint-framed record, minimal-varint length check
 *   aggregator/kafka/message_set.go:14-86, message.go:64-146   legacy sets (decoded; reaching one in
 *                                                   decodeKafkaPayload dereferences a nil RecordBatch => panic)
 *   aggregator/kafka/real_decoder.go               primitive getters incl. "insufficient data" vs other errors
 *   aggregator/kafka/length_field.go, crc32_field.go
 * Decompression is delegated by the reference to third-party modules that are NOT under /root/reference:
 *   github.com/klauspost/compress v1.16.5 (gzip, zstd), github.com/eapache/go-xerial-snappy
 *   v0.0.0-20230111030713-bf00bc1b83b6 over github.com/golang/snappy v0.0.4, github.com/pierrec/lz4/v4 v4.1.18.
 * Restated from the published formats: RFC 1952 via zlib's inflate, the snappy format description + xerial
 * framing, the LZ4 frame/block format with xxHash32 checksums, zstd via the system libzstd (dlopen; absent
 * library => decode error).
 *
 * Pinned by: CRC-32C / xxHash32 / snappy / LZ4 published check values and payloads compressed by an
 * independent implementation (pyarrow codecs) in tests/test_kafka.py.  The reference holds no test for this
 * path: parity unpinned by reference tests.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "sg_oracle.h"

enum { K_OK = 0, K_INSUF = 1, K_ERR = 2, K_PANIC = 3 };

/* ------------------------------------------------------------------ checksums */
static uint32_t crc_table[2][256]; static int crc_ready = 0;
static void crc_init(void) {
    if (crc_ready) return;
    const uint32_t poly[2] = {0xEDB88320u, 0x82F63B78u};            /* IEEE, Castagnoli (reflected) */
    for (int t = 0; t < 2; t++)
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ poly[t] : c >> 1; crc_table[t][i] = c; }
    crc_ready = 1;
}
uint32_t or_crc32(int castagnoli, const uint8_t* p, size_t n) {
    crc_init(); uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = crc_table[castagnoli ? 1 : 0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

static uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t or_xxh32(const uint8_t* p, size_t n, uint32_t seed) {
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t* end = p + n; uint32_t h;
    if (n >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        while (p + 16 <= end) {
            v1 = rotl32(v1 + rd32le(p) * P2, 13) * P1; v2 = rotl32(v2 + rd32le(p + 4) * P2, 13) * P1;
            v3 = rotl32(v3 + rd32le(p + 8) * P2, 13) * P1; v4 = rotl32(v4 + rd32le(p + 12) * P2, 13) * P1; p += 16;
        }
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else h = seed + P5;
    h += (uint32_t)n;
    while (p + 4 <= end) { h = rotl32(h + rd32le(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (*p++) * P5, 11) * P1; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

/* ------------------------------------------------------------------ growable byte buffer */
typedef struct { uint8_t* p; size_t n, cap; int is_nil; } bytes;
static int b_reserve(bytes* b, size_t more) {
    if (b->n + more <= b->cap) return 0;
    size_t nc = b->cap ? b->cap : 256; while (nc < b->n + more) nc *= 2;
    if (nc > ((size_t)1 << 30)) return -1;                          /* bound for hostile length fields */
    b->p = realloc(b->p, nc); b->cap = nc; return 0;
}
static int b_append(bytes* b, const uint8_t* s, size_t n) { if (b_reserve(b, n)) return -1; if (n) memcpy(b->p + b->n, s, n); b->n += n; return 0; }

/* ------------------------------------------------------------------ decompressors */
static int gunzip(const uint8_t* src, size_t n, bytes* out) {       /* RFC 1952, concatenated members allowed */
    z_stream z; memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) return K_ERR;
    z.next_in = (Bytef*)src; z.avail_in = (uInt)n;
    int rc = K_OK, members = 0;
    for (;;) {
        if (b_reserve(out, 4096)) { rc = K_ERR; break; }
        z.next_out = out->p + out->n; z.avail_out = (uInt)(out->cap - out->n);
        const int r = inflate(&z, Z_NO_FLUSH);
        out->n = out->cap - z.avail_out;
        if (r == Z_STREAM_END) { members++; if (z.avail_in == 0) break; if (inflateReset(&z) != Z_OK) { rc = K_ERR; break; } continue; }
        if (r != Z_OK) { rc = K_ERR; break; }
        if (z.avail_in == 0 && z.avail_out != 0) { rc = K_ERR; break; }   /* truncated member: unexpected EOF */
    }
    if (rc == K_OK && members == 0) rc = K_ERR;                     /* empty input: gzip.NewReader fails with EOF */
    inflateEnd(&z);
    return rc;
}

/* golang/snappy Decode: uvarint length, then literal / copy elements */
static int snappy_raw(const uint8_t* s, size_t n, bytes* out, int* result_nil) {
    uint64_t dlen = 0; unsigned sh = 0; size_t i = 0;
    for (;; ) {
        if (i >= n || sh > 63) return K_ERR;
        const uint8_t b = s[i++]; dlen |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) break; sh += 7;
    }
    if (dlen > 0xFFFFFFFFu) return K_ERR;                           /* ErrTooLarge */
    const size_t base = out->n;
    if (b_reserve(out, (size_t)dlen)) return K_ERR;
    size_t d = 0;
    while (i < n) {
        const uint8_t tag = s[i]; size_t len, off;
        switch (tag & 3) {
        case 0: {
            size_t x = tag >> 2;
            if (x < 60) i += 1;
            else { const size_t extra = x - 59; if (i + 1 + extra > n) return K_ERR; x = 0; for (size_t k = 0; k < extra; k++) x |= (size_t)s[i + 1 + k] << (8 * k); i += 1 + extra; }
            len = x + 1;
            if (len > dlen - d || len > n - i) return K_ERR;
            memcpy(out->p + base + d, s + i, len); d += len; i += len;
            continue;
        }
        case 1: if (i + 2 > n) return K_ERR; len = 4 + ((tag >> 2) & 7); off = ((size_t)(tag & 0xE0) << 3) | s[i + 1]; i += 2; break;
        case 2: if (i + 3 > n) return K_ERR; len = 1 + (tag >> 2); off = (size_t)s[i + 1] | (size_t)s[i + 2] << 8; i += 3; break;
        default: if (i + 5 > n) return K_ERR; len = 1 + (tag >> 2); off = rd32le(s + i + 1); i += 5; break;
        }
        if (off == 0 || off > d || len > dlen - d) return K_ERR;
        for (size_t k = 0; k < len; k++) out->p[base + d + k] = out->p[base + d + k - off];
        d += len;
    }
    if (d != dlen) return K_ERR;
    out->n = base + d;
    if (result_nil) *result_nil = dlen == 0;                         /* Decode(nil, src) of an empty block returns a nil slice */
    return K_OK;
}
/* eapache/go-xerial-snappy DecodeInto(nil, src) */
static int snappy_xerial(const uint8_t* s, size_t n, bytes* out) {
    static const uint8_t hdr[8] = {130, 83, 78, 65, 80, 80, 89, 0};
    if (n < 8) return K_ERR;
    if (memcmp(s, hdr, 8) != 0) { int nil = 0; int r = snappy_raw(s, n, out, &nil); if (r == K_OK) out->is_nil = nil; return r; }
    if (n < 20) return K_ERR;
    size_t pos = 16;
    while (pos + 4 <= n) {
        const size_t size = (size_t)s[pos] << 24 | (size_t)s[pos + 1] << 16 | (size_t)s[pos + 2] << 8 | s[pos + 3];
        pos += 4;
        if (size > n - pos) return K_ERR;
        if (snappy_raw(s + pos, size, out, NULL) != K_OK) return K_ERR;
        pos += size;
    }
    return K_OK;
}

/* LZ4 block */
static int lz4_block(const uint8_t* s, size_t n, bytes* out, size_t window_base) {
    size_t i = 0;
    while (i < n) {
        const uint8_t tok = s[i++];
        size_t lit = tok >> 4;
        if (lit == 15) { uint8_t b; do { if (i >= n) return K_ERR; b = s[i++]; lit += b; } while (b == 255); }
        if (lit > n - i) return K_ERR;
        if (b_append(out, s + i, lit)) return K_ERR;
        i += lit;
        if (i == n) return K_OK;                                    /* last sequence: literals only */
        if (i + 2 > n) return K_ERR;
        const size_t off = (size_t)s[i] | (size_t)s[i + 1] << 8; i += 2;
        size_t ml = tok & 15;
        if (ml == 15) { uint8_t b; do { if (i >= n) return K_ERR; b = s[i++]; ml += b; } while (b == 255); }
        ml += 4;
        if (off == 0 || off > out->n - window_base) return K_ERR;
        if (b_reserve(out, ml)) return K_ERR;
        for (size_t k = 0; k < ml; k++) out->p[out->n + k] = out->p[out->n + k - off];
        out->n += ml;
    }
    return K_OK;
}
/* LZ4 frame(s) as pierrec/lz4 v4 Reader consumes them: header checksum, block checksums, content checksum verified */
static int lz4_frames(const uint8_t* s, size_t n, bytes* out) {
    size_t i = 0;
    while (i < n) {
        if (n - i < 4) return K_ERR;
        const uint32_t magic = rd32le(s + i); i += 4;
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                  /* skippable frame */
            if (n - i < 4) return K_ERR;
            const uint32_t sz = rd32le(s + i); i += 4;
            if (sz > n - i) return K_ERR;
            i += sz; continue;
        }
        if (magic != 0x184D2204u) return K_ERR;
        if (n - i < 3) return K_ERR;
        const size_t desc = i;
        const uint8_t flg = s[i], bd = s[i + 1]; i += 2;
        if ((flg >> 6) != 1) return K_ERR;                           /* version */
        const int block_indep = (flg >> 5) & 1, block_sum = (flg >> 4) & 1, has_size = (flg >> 3) & 1, content_sum = (flg >> 2) & 1, dict = flg & 1;
        const unsigned bmax = (bd >> 4) & 7;
        if (bmax < 4) return K_ERR;
        const size_t block_max = (size_t)1 << (8 + 2 * bmax);
        uint64_t content_size = 0;
        if (has_size) { if (n - i < 8) return K_ERR; content_size = (uint64_t)rd32le(s + i) | (uint64_t)rd32le(s + i + 4) << 32; i += 8; }
        if (dict) { if (n - i < 4) return K_ERR; i += 4; }
        if (n - i < 1) return K_ERR;
        if (s[i] != (uint8_t)((or_xxh32(s + desc, i - desc, 0) >> 8) & 0xFF)) return K_ERR;
        i++;
        const size_t frame_base = out->n;
        for (;;) {
            if (n - i < 4) return K_ERR;
            uint32_t bs = rd32le(s + i); i += 4;
            if (bs == 0) break;                                      /* EndMark */
            const int raw = (bs >> 31) & 1; bs &= 0x7FFFFFFFu;
            if (bs > block_max || bs > n - i) return K_ERR;
            if (raw) { if (b_append(out, s + i, bs)) return K_ERR; }
            else {
                const size_t before = out->n;
                if (lz4_block(s + i, bs, out, block_indep ? before : frame_base) != K_OK) return K_ERR;
                if (out->n - before > block_max) return K_ERR;
            }
            i += bs;
            if (block_sum) { if (n - i < 4) return K_ERR; if (rd32le(s + i) != or_xxh32(s + i - bs, bs, 0)) return K_ERR; i += 4; }
        }
        if (content_sum) { if (n - i < 4) return K_ERR; if (rd32le(s + i) != or_xxh32(out->p + frame_base, out->n - frame_base, 0)) return K_ERR; i += 4; }
        if (has_size && content_size != (uint64_t)(out->n - frame_base)) return K_ERR;
    }
    return K_OK;
}

/* zstd through the system library (stable streaming ABI of libzstd >= 1.3) */
typedef struct { const void* src; size_t size, pos; } zin; typedef struct { void* dst; size_t size, pos; } zout;
static int unzstd(const uint8_t* s, size_t n, bytes* out) {
    static void* lib; static void* (*mk)(void); static size_t (*fr)(void*); static size_t (*dec)(void*, zout*, zin*); static unsigned (*iserr)(size_t); static int tried;
    if (!tried) {
        tried = 1; lib = dlopen("libzstd.so.1", RTLD_NOW);
        if (lib) { mk = (void* (*)(void))dlsym(lib, "ZSTD_createDStream"); fr = (size_t (*)(void*))dlsym(lib, "ZSTD_freeDStream");
                   dec = (size_t (*)(void*, zout*, zin*))dlsym(lib, "ZSTD_decompressStream"); iserr = (unsigned (*)(size_t))dlsym(lib, "ZSTD_isError"); }
    }
    if (!lib || !mk || !fr || !dec || !iserr) return K_ERR;
    void* ds = mk(); if (!ds) return K_ERR;
    zin in = {s, n, 0}; int rc = K_OK; size_t last = 0;
    while (in.pos < in.size) {
        if (b_reserve(out, 65536)) { rc = K_ERR; break; }
        zout o = {out->p + out->n, out->cap - out->n, 0};
        last = dec(ds, &o, &in);
        out->n += o.pos;
        if (iserr(last)) { rc = K_ERR; break; }
    }
    if (rc == K_OK && last != 0) rc = K_ERR;                         /* input ended inside a frame */
    fr(ds);
    return rc;
}

/* kafka.decompress (decompress.go:40-98).  codec None hands the input back. */
static int decompress(int codec, const uint8_t* s, size_t n, bytes* out) {
    switch (codec) {
    case 0: return b_append(out, s, n) ? K_ERR : K_OK;
    case 1: return gunzip(s, n, out);
    case 2: return snappy_xerial(s, n, out);
    case 3: return lz4_frames(s, n, out);
    case 4: return unzstd(s, n, out);
    default: return K_ERR;
    }
}
int or_kafka_decompress(int codec, const uint8_t* src, size_t n, uint8_t** out, size_t* out_n) {
    bytes b = {0, 0, 0, 0};
    const int r = decompress(codec, src, n, &b);
    if (r != K_OK) { free(b.p); *out = NULL; *out_n = 0; return -1; }
    *out = b.p ? b.p : malloc(1); *out_n = b.n; return 0;
}
void or_kafka_free(void* p) { free(p); }

/* ------------------------------------------------------------------ realDecoder */
typedef struct { const uint8_t* raw; long len, off; } rdec;
static long rem(const rdec* d) { return d->len - d->off; }
#define NEED(d, k) do { if (rem(d) < (k)) { (d)->off = (d)->len; return K_INSUF; } } while (0)
static int get_i8(rdec* d, int8_t* v) { NEED(d, 1); *v = (int8_t)d->raw[d->off]; d->off += 1; return K_OK; }
static int get_i16(rdec* d, int16_t* v) { NEED(d, 2); *v = (int16_t)((uint16_t)d->raw[d->off] << 8 | d->raw[d->off + 1]); d->off += 2; return K_OK; }
static int get_i32(rdec* d, int32_t* v) { NEED(d, 4); const uint8_t* p = d->raw + d->off; *v = (int32_t)((uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]); d->off += 4; return K_OK; }
static int get_i64(rdec* d, int64_t* v) { NEED(d, 8); uint64_t x = 0; for (int k = 0; k < 8; k++) x = x << 8 | d->raw[d->off + k]; *v = (int64_t)x; d->off += 8; return K_OK; }
/* encoding/binary.Uvarint + realDecoder.getUVarint / getVarint */
static int get_uvarint(rdec* d, uint64_t* v) {
    uint64_t x = 0; unsigned s = 0;
    for (long i = 0; d->off + i < d->len; i++) {
        const uint8_t b = d->raw[d->off + i];
        if (i == 10) { d->off += i + 1; return K_ERR; }
        if (b < 0x80) { if (i == 9 && b > 1) { d->off += i + 1; return K_ERR; } *v = x | (uint64_t)b << s; d->off += i + 1; return K_OK; }
        x |= (uint64_t)(b & 0x7F) << s; s += 7;
    }
    d->off = d->len; return K_INSUF;
}
static int get_varint(rdec* d, int64_t* v) { uint64_t u; int r = get_uvarint(d, &u); if (r != K_OK) return r; int64_t x = (int64_t)(u >> 1); if (u & 1) x = ~x; *v = x; return K_OK; }
static int get_array_len(rdec* d, long* n) {
    int32_t t; NEED(d, 4); get_i32(d, &t);
    if (t > rem(d)) { d->off = d->len; return K_INSUF; }
    if (t > 2 * 65535) return K_ERR;
    *n = t; return K_OK;
}
static int get_raw(rdec* d, long n, const uint8_t** p) {
    if (n < 0) return K_ERR;
    if (n > rem(d)) { d->off = d->len; return K_INSUF; }
    *p = d->raw + d->off; d->off += n; return K_OK;
}
/* getString / getNullableString: -1 => "" / nil */
static int get_string(rdec* d, const uint8_t** p, long* n) {
    int16_t l; int r = get_i16(d, &l); if (r != K_OK) return r;
    if (l < -1) return K_ERR;
    if (l > rem(d)) { d->off = d->len; return K_INSUF; }
    if (l == -1) { *p = d->raw + d->off; *n = 0; return K_OK; }
    *p = d->raw + d->off; *n = l; d->off += l; return K_OK;
}
static int get_bytes32(rdec* d, const uint8_t** p, long* n, int* is_nil) {   /* getBytes */
    int32_t l; int r = get_i32(d, &l); if (r != K_OK) return r;
    if (l == -1) { *is_nil = 1; *p = NULL; *n = 0; return K_OK; }
    *is_nil = 0; *n = l; return get_raw(d, l, p);
}
static int get_varint_bytes(rdec* d, const uint8_t** p, long* n) {
    int64_t l; int r = get_varint(d, &l); if (r != K_OK) return r;
    if (l == -1) { *p = NULL; *n = 0; return K_OK; }
    if (l < 0) return K_ERR;
    if (l > (int64_t)rem(d)) { d->off = d->len; return K_INSUF; }
    *n = (long)l; return get_raw(d, (long)l, p);
}
static int varint_size(int64_t v) { uint64_t u = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); int n = 1; while (u >= 0x80) { u >>= 7; n++; } return n; }

/* ------------------------------------------------------------------ result */
typedef struct { uint8_t* topic; size_t topic_n; int32_t partition; uint8_t* key; size_t key_n; uint8_t* value; size_t value_n; } kmsg;
struct or_kafka_result { kmsg* m; size_t n, cap; int status; };

typedef struct { uint8_t* key; size_t key_n; uint8_t* value; size_t value_n; } krec;
typedef struct { int32_t id; int legacy; int nil_records; krec* recs; size_t n, cap; } kpart;    /* one Records (or RecordsSet) */
typedef struct { uint8_t* name; size_t name_n; kpart* parts; size_t n, cap; } ktopic;
typedef struct { ktopic* t; size_t n, cap; } kmap;

static void part_clear(kpart* p) { for (size_t i = 0; i < p->n; i++) { free(p->recs[i].key); free(p->recs[i].value); } free(p->recs); p->recs = NULL; p->n = p->cap = 0; p->legacy = 0; p->nil_records = 0; }
static void topic_clear(ktopic* t) { for (size_t i = 0; i < t->n; i++) part_clear(&t->parts[i]); free(t->parts); t->parts = NULL; t->n = t->cap = 0; }
static void map_free(kmap* m) { for (size_t i = 0; i < m->n; i++) { topic_clear(&m->t[i]); free(m->t[i].name); } free(m->t); }
static uint8_t* dupb(const uint8_t* p, size_t n) { uint8_t* r = malloc(n ? n : 1); if (n) memcpy(r, p, n); return r; }
/* r.Records[topic] = make(map...): an existing topic loses its partitions */
static ktopic* map_topic(kmap* m, const uint8_t* name, size_t n) {
    for (size_t i = 0; i < m->n; i++) if (m->t[i].name_n == n && memcmp(m->t[i].name, name, n) == 0) { topic_clear(&m->t[i]); return &m->t[i]; }
    if (m->n == m->cap) { m->cap = m->cap ? m->cap * 2 : 4; m->t = realloc(m->t, m->cap * sizeof(ktopic)); }
    ktopic* t = &m->t[m->n++]; memset(t, 0, sizeof *t); t->name = dupb(name, n); t->name_n = n; return t;
}
/* r.Records[topic][partition] = records: a repeated partition is overwritten */
static kpart* topic_part(ktopic* t, int32_t id) {
    for (size_t i = 0; i < t->n; i++) if (t->parts[i].id == id) { part_clear(&t->parts[i]); return &t->parts[i]; }
    if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 4; t->parts = realloc(t->parts, t->cap * sizeof(kpart)); }
    kpart* p = &t->parts[t->n++]; memset(p, 0, sizeof *p); p->id = id; return p;
}
static void part_add(kpart* p, const uint8_t* k, long kn, const uint8_t* v, long vn) {
    if (p->n == p->cap) { p->cap = p->cap ? p->cap * 2 : 8; p->recs = realloc(p->recs, p->cap * sizeof(krec)); }
    krec* r = &p->recs[p->n++]; r->key = dupb(k, (size_t)kn); r->key_n = (size_t)kn; r->value = dupb(v, (size_t)vn); r->value_n = (size_t)vn;
}

/* ------------------------------------------------------------------ Record / RecordBatch / MessageSet */
/* Record.decode (record.go:41-87) */
static int record_decode(rdec* d, kpart* out) {
    const long start = d->off; int64_t length; int r = get_varint(d, &length); if (r != K_OK) return r;
    int8_t attr; int64_t ts, od, nh; const uint8_t *k, *v; long kn, vn;
    if ((r = get_i8(d, &attr)) != K_OK) return r;
    if ((r = get_varint(d, &ts)) != K_OK) return r;
    if ((r = get_varint(d, &od)) != K_OK) return r;
    if ((r = get_varint_bytes(d, &k, &kn)) != K_OK) return r;
    if ((r = get_varint_bytes(d, &v, &vn)) != K_OK) return r;
    if ((r = get_varint(d, &nh)) != K_OK) return r;
    if (nh > ((int64_t)1 << 45)) return K_PANIC;                     /* make([]*RecordHeader, n): len out of range */
    for (int64_t i = 0; i < nh; i++) {
        const uint8_t* p; long n;
        if ((r = get_varint_bytes(d, &p, &n)) != K_OK) return r;
        if ((r = get_varint_bytes(d, &p, &n)) != K_OK) return r;
    }
    if ((int64_t)(d->off - start - varint_size(length)) != length) return K_ERR;   /* varintLengthField.check */
    part_add(out, k, kn, v, vn);
    return K_OK;
}

/* RecordBatch.decode (record_batch.go:51-139).  *partial: PartialTrailingRecord */
static int record_batch_decode(rdec* d, kpart* out, int* partial, size_t* n_records) {
    int64_t i64; int32_t batch_len, i32; int8_t ver; int16_t attrs, i16; int r;
    *partial = 0; *n_records = 0;
    if ((r = get_i64(d, &i64)) != K_OK) return r;
    if ((r = get_i32(d, &batch_len)) != K_OK) return r;
    if ((r = get_i32(d, &i32)) != K_OK) return r;
    if ((r = get_i8(d, &ver)) != K_OK) return r;
    const long crc_at = d->off; NEED(d, 4); d->off += 4;             /* push(crc32 castagnoli) */
    if ((r = get_i16(d, &attrs)) != K_OK) return r;
    if ((r = get_i32(d, &i32)) != K_OK) return r;
    if ((r = get_i64(d, &i64)) != K_OK) return r;
    if ((r = get_i64(d, &i64)) != K_OK) return r;
    if ((r = get_i64(d, &i64)) != K_OK) return r;
    if ((r = get_i16(d, &i16)) != K_OK) return r;
    if ((r = get_i32(d, &i32)) != K_OK) return r;
    long num; if ((r = get_array_len(d, &num)) != K_OK) return r;
    const uint8_t* rec; r = get_raw(d, (long)batch_len - 49, &rec);
    if (r == K_INSUF) { *partial = 1; return K_OK; }
    if (r != K_OK) return r;
    const long rec_n = (long)batch_len - 49;
    const uint32_t want = (uint32_t)d->raw[crc_at] << 24 | (uint32_t)d->raw[crc_at + 1] << 16 | (uint32_t)d->raw[crc_at + 2] << 8 | d->raw[crc_at + 3];
    if (or_crc32(1, d->raw + crc_at + 4, (size_t)(d->off - crc_at - 4)) != want) return K_ERR;
    bytes plain = {0, 0, 0, 0};
    const int codec = (int8_t)attrs & 7;
    if (decompress(codec, rec, (size_t)rec_n, &plain) != K_OK) { free(plain.p); return K_ERR; }
    if (plain.is_nil) {                                              /* decode(nil, ...) returns at once: Records keeps `num` nil entries */
        free(plain.p);
        if (num > 0) { out->nil_records = 1; *n_records = (size_t)num; }
        return K_OK;
    }
    rdec rd = {plain.p, (long)plain.n, 0};
    const size_t before = out->n;
    for (long i = 0; i < num; i++) {
        r = record_decode(&rd, out);
        if (r != K_OK) break;
    }
    if (r == K_OK && rd.off != rd.len) r = K_ERR;                    /* decode(): "invalid length" (also when numRecs < 0) */
    if (r == K_INSUF) {                                              /* PartialTrailingRecord: Records = nil */
        while (out->n > before) { out->n--; free(out->recs[out->n].key); free(out->recs[out->n].value); }
        *partial = 1; r = K_OK;
    } else if (r != K_OK) {
        while (out->n > before) { out->n--; free(out->recs[out->n].key); free(out->recs[out->n].value); }
    } else *n_records = out->n - before;
    free(plain.p);
    return r;
}

static int message_set_decode(rdec* d, size_t* n_msgs, int* partial, int* overflow, int depth);
/* Message.decode (message.go:64-140) inside MessageBlock.decode (message_set.go:14-41) */
static int message_block_decode(rdec* d, int64_t* offset, int depth) {
    int r; if ((r = get_i64(d, offset)) != K_OK) return r;
    int32_t length; const long len_at = d->off;
    if ((r = get_i32(d, &length)) != K_OK) return r;                 /* lengthField.decode */
    if (length > (int32_t)rem(d)) return K_INSUF;
    const long crc_at = d->off; NEED(d, 4); d->off += 4;             /* push(crc32 IEEE) */
    int8_t ver, attr; int64_t ts;
    if ((r = get_i8(d, &ver)) != K_OK) return r;
    if (ver > 1) return K_ERR;
    if ((r = get_i8(d, &attr)) != K_OK) return r;
    if (ver == 1 && (r = get_i64(d, &ts)) != K_OK) return r;
    const uint8_t *k, *v; long kn, vn; int knil, vnil;
    if ((r = get_bytes32(d, &k, &kn, &knil)) != K_OK) return r;
    if ((r = get_bytes32(d, &v, &vn, &vnil)) != K_OK) return r;
    const int codec = attr & 7;
    if (!vnil && codec != 0) {
        bytes plain = {0, 0, 0, 0};
        if (decompress(codec, v, (size_t)vn, &plain) != K_OK) { free(plain.p); return K_ERR; }
        if (depth > 8) { free(plain.p); return K_ERR; }
        rdec in = {plain.p, (long)plain.n, 0}; size_t nm; int pa, ov;
        r = message_set_decode(&in, &nm, &pa, &ov, depth + 1);
        free(plain.p);
        if (r != K_OK) return r;
    }
    const uint32_t want = (uint32_t)d->raw[crc_at] << 24 | (uint32_t)d->raw[crc_at + 1] << 16 | (uint32_t)d->raw[crc_at + 2] << 8 | d->raw[crc_at + 3];
    if (or_crc32(0, d->raw + crc_at + 4, (size_t)(d->off - crc_at - 4)) != want) return K_ERR;
    if ((int32_t)(d->off - len_at - 4) != length) return K_ERR;      /* lengthField.check */
    return K_OK;
}
/* MessageSet.decode (message_set.go:49-86) */
static int message_set_decode(rdec* d, size_t* n_msgs, int* partial, int* overflow, int depth) {
    *n_msgs = 0; *partial = 0; *overflow = 0;
    while (rem(d) > 0) {
        if (rem(d) < 17) { *partial = 1; return K_OK; }              /* magicValue: peekInt8(16) */
        if ((int8_t)d->raw[d->off + 16] > 1) return K_OK;
        int64_t offset = 0; const int r = message_block_decode(d, &offset, depth);
        if (r == K_OK) (*n_msgs)++;
        else if (r == K_INSUF) { if (offset == -1) *overflow = 1; else *partial = 1; return K_OK; }
        else return r;
    }
    return K_OK;
}

/* Records.decode (records.go:46-71) into one partition slot; returns what numRecords()/isPartial()/isOverflow() would */
static int records_decode(rdec* d, kpart* out, size_t* n, int* partial, int* overflow, int* legacy) {
    *n = 0; *partial = 0; *overflow = 0; *legacy = 0;
    if (rem(d) < 17) return K_INSUF;                                 /* setTypeFromMagic */
    if ((int8_t)d->raw[d->off + 16] < 2) { *legacy = 1; return message_set_decode(d, n, partial, overflow, 0); }
    return record_batch_decode(d, out, partial, n);
}

/* ------------------------------------------------------------------ Produce request / Fetch response */
static int produce_decode(const uint8_t* payload, size_t size, kmap* m) {
    if (size < 4) return K_ERR;                                      /* io.ReadFull of the length */
    const int32_t length = (int32_t)((uint32_t)payload[0] << 24 | (uint32_t)payload[1] << 16 | (uint32_t)payload[2] << 8 | payload[3]);
    if (length <= 4 || length > 100 * 1024 * 1024) return K_ERR;
    if ((size_t)length > size - 4) return K_ERR;                     /* io.ReadFull of the body: unexpected EOF */
    rdec d = {payload + 4, length, 0};
    int16_t key, version; int32_t corr, timeout; int16_t acks; const uint8_t* s; long sn; int r;
    if ((r = get_i16(&d, &key)) != K_OK) return r;
    if ((r = get_i16(&d, &version)) != K_OK) return r;
    if ((r = get_i32(&d, &corr)) != K_OK) return r;
    if ((r = get_string(&d, &s, &sn)) != K_OK) return r;
    if (key != 0) return K_ERR;                                      /* allocateBody: only Produce is known */
    if (version >= 3 && (r = get_string(&d, &s, &sn)) != K_OK) return r;   /* transactional id */
    if ((r = get_i16(&d, &acks)) != K_OK) return r;
    if ((r = get_i32(&d, &timeout)) != K_OK) return r;
    long topics; if ((r = get_array_len(&d, &topics)) != K_OK) return r;
    for (long i = 0; i < topics; i++) {                              /* topicCount 0 (and -1) leave no records */
        if ((r = get_string(&d, &s, &sn)) != K_OK) return r;
        long parts; if ((r = get_array_len(&d, &parts)) != K_OK) return r;
        ktopic* t = map_topic(m, s, (size_t)sn);
        for (long j = 0; j < parts; j++) {
            int32_t id, sz; const uint8_t* sub;
            if ((r = get_i32(&d, &id)) != K_OK) return r;
            if ((r = get_i32(&d, &sz)) != K_OK) return r;
            if ((r = get_raw(&d, sz, &sub)) != K_OK) return r;
            rdec rd = {sub, sz, 0}; kpart tmp; memset(&tmp, 0, sizeof tmp); size_t n; int pa, ov, legacy;
            r = records_decode(&rd, &tmp, &n, &pa, &ov, &legacy);
            if (r != K_OK) { part_clear(&tmp); return r; }
            kpart* p = topic_part(t, id);
            p->recs = tmp.recs; p->n = tmp.n; p->cap = tmp.cap; p->legacy = legacy; p->nil_records = tmp.nil_records;
        }
    }
    if (d.off != d.len) return K_ERR;                                /* decode(): "invalid length" */
    return K_OK;
}

static int fetch_decode(const uint8_t* payload, size_t size, int16_t version, kmap* m) {
    rdec d = {payload, (long)size, 0};
    int32_t length, corr; int r;
    if ((r = get_i32(&d, &length)) != K_OK) return r;                /* ResponseHeader.decode */
    if (length <= 4 || length > 100 * 1024 * 1024) return K_ERR;
    r = get_i32(&d, &corr);
    if (version >= 12) {                                             /* header v1: tagged fields */
        uint64_t tags, x;
        int r2; if ((r2 = get_uvarint(&d, &tags)) != K_OK) return r2;
        for (uint64_t i = 0; i < tags; i++) {
            const uint8_t* p;
            if ((r2 = get_uvarint(&d, &x)) != K_OK) return r2;
            if ((r2 = get_uvarint(&d, &x)) != K_OK) return r2;
            if (x > 0x7FFFFFFF) return K_ERR;
            if ((r2 = get_raw(&d, (long)x, &p)) != K_OK) return r2;
        }
    }
    if (r != K_OK) return r;
    rdec b = {payload + d.off, (long)size - d.off, 0};               /* payload = payload[off:] */
    int32_t i32; int16_t i16; int64_t i64;
    if (version >= 1 && (r = get_i32(&b, &i32)) != K_OK) return r;
    if (version >= 7) { if ((r = get_i16(&b, &i16)) != K_OK) return r; if ((r = get_i32(&b, &i32)) != K_OK) return r; }
    long topics; if ((r = get_array_len(&b, &topics)) != K_OK) return r;
    /* negative counts pass getArrayLength; make(map, n<0) is legal and the loops simply do not run */
    for (long i = 0; i < topics; i++) {
        const uint8_t* s; long sn;
        if ((r = get_string(&b, &s, &sn)) != K_OK) return r;
        long blocks; if ((r = get_array_len(&b, &blocks)) != K_OK) return r;
        ktopic* t = map_topic(m, s, (size_t)sn);
        for (long j = 0; j < blocks; j++) {
            int32_t id; if ((r = get_i32(&b, &id)) != K_OK) return r;
            /* FetchResponseBlock.decode */
            if ((r = get_i16(&b, &i16)) != K_OK) return r;
            if ((r = get_i64(&b, &i64)) != K_OK) return r;
            if (version >= 4) {
                if ((r = get_i64(&b, &i64)) != K_OK) return r;
                if (version >= 5 && (r = get_i64(&b, &i64)) != K_OK) return r;
                long nt; if ((r = get_array_len(&b, &nt)) != K_OK) return r;
                for (long k = 0; k < nt; k++) { if ((r = get_i64(&b, &i64)) != K_OK) return r; if ((r = get_i64(&b, &i64)) != K_OK) return r; }
            }
            if (version >= 11 && (r = get_i32(&b, &i32)) != K_OK) return r;
            int32_t rsz; const uint8_t* sub;
            if ((r = get_i32(&b, &rsz)) != K_OK) return r;
            if ((r = get_raw(&b, rsz, &sub)) != K_OK) return r;
            rdec rd = {sub, rsz, 0};
            kpart acc; memset(&acc, 0, sizeof acc); int sets = 0;
            while (rem(&rd) > 0) {
                kpart tmp; memset(&tmp, 0, sizeof tmp); size_t n; int pa, ov, legacy;
                r = records_decode(&rd, &tmp, &n, &pa, &ov, &legacy);
                if (r == K_INSUF) { part_clear(&tmp); break; }
                if (r != K_OK) { part_clear(&tmp); part_clear(&acc); return r; }
                if (n > 0 || (pa && sets == 0)) {                    /* appended to RecordsSet */
                    sets++;
                    if (legacy) acc.legacy = 1;                       /* record.RecordBatch is nil there */
                    if (tmp.nil_records) acc.nil_records = 1;
                    for (size_t q = 0; q < tmp.n; q++) { part_add(&acc, tmp.recs[q].key, (long)tmp.recs[q].key_n, tmp.recs[q].value, (long)tmp.recs[q].value_n); }
                }
                part_clear(&tmp);
                if (pa || ov) break;
            }
            kpart* p = topic_part(t, id);
            p->recs = acc.recs; p->n = acc.n; p->cap = acc.cap; p->legacy = acc.legacy; p->nil_records = acc.nil_records;
        }
    }
    return K_OK;
}

static int cmp_topic(const void* a, const void* b) {
    const ktopic *x = a, *y = b; const size_t n = x->name_n < y->name_n ? x->name_n : y->name_n;
    const int c = memcmp(x->name, y->name, n); if (c) return c;
    return (x->name_n > y->name_n) - (x->name_n < y->name_n);
}
static int cmp_part(const void* a, const void* b) { const kpart *x = a, *y = b; return (x->id > y->id) - (x->id < y->id); }

/* decodeKafkaPayload.  method_id 1 = PRODUCE_REQUEST, 2 = FETCH_RESPONSE.  Messages come out ordered by
 * (topic, partition, position) — the reference iterates Go maps, i.e. in no defined order. */
or_kafka_result* or_kafka_decode(const uint8_t* payload, size_t size, int method_id, int16_t api_version) {
    or_kafka_result* res = calloc(1, sizeof *res);
    kmap m = {0, 0, 0}; int r = K_OK;
    if (method_id == 1) r = produce_decode(payload, size, &m);
    else if (method_id == 2) r = fetch_decode(payload, size, api_version, &m);
    else { res->status = 0; return res; }                            /* neither branch: empty result, no error */
    if (r == K_OK)
        for (size_t i = 0; i < m.n && r == K_OK; i++)
            for (size_t j = 0; j < m.t[i].n; j++)
                if (m.t[i].parts[j].legacy || m.t[i].parts[j].nil_records) { r = K_PANIC; break; }   /* nil RecordBatch / nil *Record */
    res->status = r;
    if (r == K_OK) {
        if (m.n) qsort(m.t, m.n, sizeof(ktopic), cmp_topic);
        for (size_t i = 0; i < m.n; i++) {
            if (m.t[i].n) qsort(m.t[i].parts, m.t[i].n, sizeof(kpart), cmp_part);
            for (size_t j = 0; j < m.t[i].n; j++)
                for (size_t q = 0; q < m.t[i].parts[j].n; q++) {
                    if (res->n == res->cap) { res->cap = res->cap ? res->cap * 2 : 16; res->m = realloc(res->m, res->cap * sizeof(kmsg)); }
                    kmsg* k = &res->m[res->n++]; const krec* rc = &m.t[i].parts[j].recs[q];
                    k->topic = dupb(m.t[i].name, m.t[i].name_n); k->topic_n = m.t[i].name_n; k->partition = m.t[i].parts[j].id;
                    k->key = dupb(rc->key, rc->key_n); k->key_n = rc->key_n; k->value = dupb(rc->value, rc->value_n); k->value_n = rc->value_n;
                }
        }
    }
    map_free(&m);
    return res;
}
void or_kafka_result_free(or_kafka_result* r) {
    if (!r) return;
    for (size_t i = 0; i < r->n; i++) { free(r->m[i].topic); free(r->m[i].key); free(r->m[i].value); }
    free(r->m); free(r);
}
size_t or_kafka_count(const or_kafka_result* r) { return r->n; }
int or_kafka_status(const or_kafka_result* r) { return r->status; }   /* 0 ok, 1 insufficient data, 2 error, 3 panic */
int or_kafka_msg(const or_kafka_result* r, size_t i, const uint8_t** topic, size_t* topic_n, int32_t* partition,
                 const uint8_t** key, size_t* key_n, const uint8_t** value, size_t* value_n) {
    if (i >= r->n) return -1;
    const kmsg* k = &r->m[i];
    *topic = k->topic; *topic_n = k->topic_n; *partition = k->partition; *key = k->key; *key_n = k->key_n; *value = k->value; *value_n = k->value_n;
    return 0;
}
