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
uint32_t or_sl_owner(const or_sockline* s, uint64_t* fd) { if (fd) *fd = s->fd; return s->pid; }
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

/* ---------------------------------------------------------------------------------------------- *
 * A new line seeded from the proc file system — NewSocketLine(fetch = true) :38-54 →
 * getConnectionInfo :399-429.  `proc_root` stands for "/proc" (the tests point it at a directory they
 * built), `now_kernel_ns` for convertUserTimeToKernelTime(time.Now()) (data.go:1745-1747).
 * ---------------------------------------------------------------------------------------------- */
#include <stdio.h>
#include <unistd.h>

/* getInodeFromFD :351-366 — regexp `socket:\[(\d+)\]`, FindStringSubmatch: the FIRST match anywhere in
 * the link text, at least one digit.  Returns 0 and the digits, or -1 ("no inode found in link"). */
int or_sl_inode_from_link(const char* link, char* inode, size_t cap) {
    static const char lead[] = "socket:[";
    for (const char* p = strstr(link, lead); p; p = strstr(p + 1, lead)) {
        const char* d = p + sizeof lead - 1; size_t n = 0;
        while (d[n] >= '0' && d[n] <= '9') n++;
        if (n > 0 && d[n] == ']') {
            if (n + 1 > cap) return -1;
            memcpy(inode, d, n); inode[n] = 0;
            return 0;
        }
    }
    return -1;
}

/* strconv.ParseInt(s, 16, 64) as the callers below use it (error ignored): the value on success; 0 on a
 * syntax error (empty, a non-hex digit, a lone sign); the clamped extreme on overflow. */
static long long parse_hex_i64(const char* s, size_t n) {
    if (n == 0) return 0;
    int neg = 0; size_t i = 0;
    if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; if (n == 1) return 0; }
    unsigned long long v = 0; int over = 0;
    for (; i < n; i++) {
        int c = (unsigned char)s[i], d;
        if (c >= '0' && c <= '9') d = c - '0'; else if (c >= 'a' && c <= 'f') d = c - 'a' + 10; else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else return 0;                                  /* (an underscore is a syntax error with an explicit base) */
        if (v >> 60) over = 1;
        v = (v << 4) | (unsigned)d;
    }
    if (over || v > (neg ? 1ull << 63 : (1ull << 63) - 1)) return neg ? (long long)(1ull << 63) : (long long)((1ull << 63) - 1);
    return neg ? -(long long)v : (long long)v;
}

/* convertHexToIP :332-340: one "%d" per two hex characters, order reversed, joined by dots */
static void hex_to_ip(const char* hex8, char out[16]) {
    long long part[4];
    for (int i = 0; i < 4; i++) part[i] = parse_hex_i64(hex8 + 2 * i, 2);
    snprintf(out, 16, "%lld.%lld.%lld.%lld", part[3], part[2], part[1], part[0]);
}

/* parseTcpLine :384-397 — strings.Fields, fields[1] / fields[2], address = [:8], port = [9:].
 * The reference indexes without checking (a short line would panic; /proc never produces one): -1 here. */
int or_sl_parse_tcp_line(const char* line, char lip[16], int* lport, char rip[16], int* rport) {
    const char* f[3]; size_t fl[3]; int nf = 0;
    const char* p = line;
    while (*p && nf < 3) {
        while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r') p++;
        if (!*p) break;
        f[nf] = p; while (*p && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r')) p++;
        fl[nf] = (size_t)(p - f[nf]); nf++;
    }
    if (nf < 3 || fl[1] < 9 || fl[2] < 9) return -1;
    hex_to_ip(f[1], lip); hex_to_ip(f[2], rip);
    long long lp = parse_hex_i64(f[1] + 9, fl[1] - 9), rp = parse_hex_i64(f[2] + 9, fl[2] - 9);   /* convertHexToPort :343-349 */
    *lport = (lp < 0 || lp > 65535) ? 0 : (int)lp;
    *rport = (rp < 0 || rp > 65535) ? 0 : (int)rp;
    return 0;
}

/* getConnectionInfo :399-429.  0 = seeded (all previous values cleared, one open value at now_kernel_ns);
 * 1 readlink failed, 2 no inode in the link, 3 net/tcp not readable, 4 no line contains the inode
 * (findTCPConnection :368-382: the first line of the file that CONTAINS the inode's digits anywhere —
 * header line included, and an address or queue column may match before the inode column does). */
int or_sl_seed_from_proc(or_sockline* s, const char* proc_root, uint64_t now_kernel_ns) {
    char path[512], link[256], inode[32];
    snprintf(path, sizeof path, "%s/%u/fd/%llu", proc_root, s->pid, (unsigned long long)s->fd);
    ssize_t ln = readlink(path, link, sizeof link - 1);
    if (ln < 0) return 1;
    link[ln] = 0;
    if (or_sl_inode_from_link(link, inode, sizeof inode) != 0) return 2;
    snprintf(path, sizeof path, "%s/%u/net/tcp", proc_root, s->pid);
    FILE* f = fopen(path, "r");
    if (!f) return 3;
    char* line = NULL; size_t cap = 0; int found = 0;
    while (getline(&line, &cap, f) >= 0) {
        size_t l = strlen(line);
        if (l > 65536) break;                                   /* bufio.Scanner gives up on a token above 64 KiB: the scan ends */
        if (l && line[l - 1] == '\n') line[--l] = 0;
        if (l && line[l - 1] == '\r') line[--l] = 0;            /* ScanLines drops one trailing \r */
        if (strstr(line, inode)) { found = 1; break; }
    }
    fclose(f);
    if (!found) { free(line); return 4; }
    or_sockinfo si; memset(&si, 0, sizeof si);
    int lp = 0, rp = 0;
    if (or_sl_parse_tcp_line(line, si.saddr, &lp, si.daddr, &rp) != 0) { free(line); return 5; }
    free(line);
    si.pid = s->pid; si.fd = s->fd; si.sport = (uint16_t)lp; si.dport = (uint16_t)rp;
    s->n = 0;                                                   /* ClearAll :56-60 */
    or_sl_add(s, now_kernel_ns, &si);
    return 0;
}
