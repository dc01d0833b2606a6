/*
 * sockline.c — CPU restatement of the reference's socket-line state machine.
 * TEST INFRASTRUCTURE ONLY (see sg_oracle.h).
 *
 * Follows aggregator/sock_num_line.go:
 *   TimestampedSocket / SocketLine          :23-36
 *   AddValue                                :62-81   (last-equal de-duplication, then sorted insert)
 *   GetValue                                :83-157  (binary search + the three special cases)
 *   DeleteUnused                            :159-208 (incl. its quirk: the trailing element of the first
 *                                                     pass is dropped when it is not the 2nd of a pair)
 *   insertIntoSortedSlice                   :311-322 (lower bound on Timestamp: equal stamps go BEFORE)
 * time.Now() is a parameter here (now_ns), so the state machine is deterministic.
 * Pinned by the reference's own known-answer tests, re-encoded in tests/test_sockline.py:
 *   aggregator/sock_line_test.go:16-347 (TestSocketLine), :443-476 (TestXxx2), :478-501
 *   (TestAlreadyEstablishCanBeFound).
 */
#include <stdlib.h>
#include <string.h>
#include "sg_oracle.h"

typedef struct {
    uint64_t ts, last_match;
    int open;                       /* SockInfo != nil */
    or_sockinfo si;
} ts_sock;

struct or_sockline {
    uint32_t pid; uint64_t fd;
    ts_sock* v; size_t n, cap;
};

or_sockline* or_sl_create(uint32_t pid, uint64_t fd) {
    or_sockline* s = calloc(1, sizeof *s);
    s->pid = pid; s->fd = fd;
    return s;
}
void or_sl_destroy(or_sockline* s) { if (s) { free(s->v); free(s); } }
size_t or_sl_len(const or_sockline* s) { return s->n; }
int or_sl_at(const or_sockline* s, size_t i, uint64_t* ts, uint64_t* last_match, or_sockinfo* si) {
    if (i >= s->n) return -1;
    if (ts) *ts = s->v[i].ts;
    if (last_match) *last_match = s->v[i].last_match;
    if (si && s->v[i].open) *si = s->v[i].si;
    return s->v[i].open;
}

static int same_pair(const or_sockinfo* a, const or_sockinfo* b) {
    return strcmp(a->saddr, b->saddr) == 0 && a->sport == b->sport && strcmp(a->daddr, b->daddr) == 0 && a->dport == b->dport;
}

/* sock_num_line.go:62-81 + :311-322.  si == NULL is a close event. */
void or_sl_add(or_sockline* s, uint64_t ts, const or_sockinfo* si) {
    if (s->n > 0) {
        const ts_sock* last = &s->v[s->n - 1];
        if (last->open && si && same_pair(&last->si, si)) return;
    }
    size_t lo = 0, hi = s->n;                                    /* sort.Search: first i with v[i].ts >= ts */
    while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (s->v[mid].ts >= ts) hi = mid; else lo = mid + 1; }
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 8; s->v = realloc(s->v, s->cap * sizeof(ts_sock)); }
    memmove(&s->v[lo + 1], &s->v[lo], (s->n - lo) * sizeof(ts_sock));
    memset(&s->v[lo], 0, sizeof(ts_sock));
    s->v[lo].ts = ts; s->v[lo].open = si != NULL;
    if (si) s->v[lo].si = *si;
    s->n++;
}

/* sock_num_line.go:83-157.  Returns OR_SL_OK and fills *out, or one of the error codes. */
int or_sl_get(or_sockline* s, uint64_t ts, uint64_t now_ns, or_sockinfo* out) {
    if (s->n == 0) return OR_SL_EMPTY;                           /* "sock line is empty" */
    size_t lo = 0, hi = s->n;                                    /* first i with !(v[i].ts < ts) */
    while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (!(s->v[mid].ts < ts)) hi = mid; else lo = mid + 1; }
    const size_t index = lo;
    if (index == s->n) {
        s->v[index - 1].last_match = now_ns;
        if (!s->v[s->n - 1].open) {
            if (index >= 2 && s->v[index - 2].open && (ts - s->v[index - 2].ts) < 60ull * 1000000000ull) {
                *out = s->v[index - 2].si;
                return OR_SL_OK;
            }
            return OR_SL_CLOSED_LAST;                            /* "closed socket on last entry" */
        }
        *out = s->v[s->n - 1].si;
        return OR_SL_OK;
    }
    if (index == 0) {
        if (s->v[0].open) { *out = s->v[0].si; return OR_SL_OK; }
        return OR_SL_NO_SMALLER;                                 /* "no smaller value found" */
    }
    if (!s->v[index - 1].open) {
        const ts_sock* prev = index >= 2 ? &s->v[index - 2] : NULL;
        const ts_sock* after = index < s->n ? &s->v[index] : NULL;
        if (prev && prev->open && after && after->open &&
            strcmp(prev->si.daddr, after->si.daddr) == 0 && prev->si.dport == after->si.dport) {
            if (ts - prev->ts < after->ts - ts) *out = prev->si; else *out = after->si;
            return OR_SL_OK;
        }
        return OR_SL_CLOSED;                                     /* "closed socket" */
    }
    s->v[index - 1].last_match = now_ns;
    *out = s->v[index - 1].si;
    return OR_SL_OK;
}

/* sock_num_line.go:159-208 */
void or_sl_delete_unused(or_sockline* s) {
    if (s->n <= 1) return;
    ts_sock* r = malloc(s->n * sizeof(ts_sock)); size_t rn = 0, i = 0;
    while (i < s->n - 1) {
        if (s->v[i].open && s->v[i + 1].open) { r[rn++] = s->v[i + 1]; i += 2; }
        else { r[rn++] = s->v[i]; i++; }
    }
    /* (an element left at i == n-1 is not appended: reference behaviour) */
    memcpy(s->v, r, rn * sizeof(ts_sock)); s->n = rn; free(r);

    uint64_t last_matched = 0;
    for (size_t k = s->n; k-- > 0;) if (s->v[k].last_match != 0 && s->v[k].last_match > last_matched) last_matched = s->v[k].last_match;
    const uint64_t assumed = 5ull * 60ull * 1000000000ull;
    for (long k = (long)s->n - 1; k >= 1; k--) {
        if (!s->v[k].open && s->v[k - 1].open && s->v[k - 1].last_match + assumed < last_matched) {
            memmove(&s->v[k - 1], &s->v[k + 1], (s->n - (size_t)k - 1) * sizeof(ts_sock));
            s->n -= 2;
            k--;
        }
    }
}
