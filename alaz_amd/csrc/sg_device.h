// sg_device.h — device-side data layout shared by the kernels and the host engine.
// gfx950 only (wave64, 160 KiB LDS/CU, 8 XCDs); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/servicegraph.h"

typedef unsigned long long u64;
typedef uint32_t u32;

#define SG_NONE       0xFFFFFFFFu
#define SG_EKEY_EMPTY (~0ull)
#define SG_WAVE       64

// Join table: bucketized cuckoo hash over BOTH reference maps (PodIPToPodUid and
// ServiceIPToServiceUid, aggregator/cluster.go:13-17).  A bucket is two 8-byte entries (one 16-byte
// load); an IP lives in bucket h1(ip) or bucket h2(ip), so a lookup is two independent loads and
// four compares — no probe loop, no divergence between the lanes of a wave.
// entry: low 32 bits = IP, high 32 bits = kind << 30 | id; kind 1 = pod, 2 = service, 3 = the IP
// is in both maps (id = the service; the pod id is in the small second table).  All ones = empty.
#define SG_IP_EMPTY   (~0ull)
#define SG_IP_LDS_MAX 4096      // entries (32 KiB): a residual cuckoo table up to this size may be staged in LDS by K1

// Join, partitioned K1 (variant 0): a two-level block table in front of the cuckoo table.  Pod CIDRs are dense per
// node (/24 per kubelet), so most of a cluster's IPs live in few /24 blocks:
//   level 1  jl1[]: 2-choice hash of the block number b = ip >> 8 -> {tag = b, blk}; u64 entries, all-ones tag = empty;
//            always staged in LDS (two independent ds_read_b64, no probe loop)
//   level 2  jl2[blk * 256 + (ip & 255)] = kind << 30 | id, 0 = no such IP; block 0 is the all-zero "null" block a miss
//            is steered to (no select on the result, no out-of-range read); staged in LDS when it fits beside the
//            edge cache (C3: 60 blocks = 60 KiB), read from global memory otherwise
// IPs of blocks that got no level-2 block (sparse /24s, block budget exhausted) stay in the cuckoo table (`iptab`,
// "residual"); the fast path of k1a_partition hands an event whose block lookup missed to the general path only when
// that table is not empty.  Variant 1 (global edge table) keeps every IP in the cuckoo table.
// (hash functions and constants: sg_hash.h, shared with the host-side table builder join_host.hpp)

// device counters (u64 each)
enum {
    C_N_KNOWN = 0,    // set by host before close
    C_N_LABELS,       // max(host-declared, max label seen)
    C_N_OBIP,         // distinct raw-IP outbound nodes of the window
    C_N_EDGES,        // edges of the window (min(found, max_edges))
    C_N_EVENTS,       // accepted events of the window
    C_DROPPED_SRC,    // window
    C_DROPPED_CAP,    // window
    C_MISROUTED,      // window
    C_TMIN_NS,        // min write_time_ns over accepted events
    C_TMAX_NS,
    C_N_NODES,        // NK + NL + NOB
    C_EDGES_FOUND,    // edges found in the table before clamping
    C_OVF_N,          // records in the partition overflow list (window)
    C_N_LONG,         // rows longer than one wave (row sort work list)
    C_HALO_OVF,       // halo requests that did not fit the per-pair capacity (must stay 0)
    C_ALIVE_N,        // SG_EV_ALIVE records accepted by K1 in the open window (list length, may exceed the capacity)
    C_ALIVE_DROP,     // ... whose endpoint was not listed at close (working counter)
    C_ALIVE_SEEN,     // closed window: records accepted
    C_ALIVE_DROPPED,  // closed window: records not marked (capacity / unlisted endpoint)
    C_ACT_L,          // world > 1: nodes whose layer output this shard computes (local sources + local leaf destinations)
    C_ACT_P,          // world > 1: nodes whose score projections this shard needs (local sources + local destinations)
    C_HUB_ITEMS,      // (row, block) work items of the rows with more than SG_MEAN_BLOCK neighbours (k2_rowptr -> k4_gather)
    // warm windows (Dev::warm): the edge set and its CSR order kept from the last cold window
    C_COLD,           // != 0: this window takes the full rebuild — 1 set by kc_prepare (no usable kept state / the host does not try), 2 by the warm pass B (a table or a partition's key budget ran full, or an edge was dropped)
    C_KEPT_VALID,     // the kept state describes a whole window (no capacity drop, no raw outbound IP when it was captured)
    C_KEPT_E,         // edges of the kept CSR
    C_KEPT_NK,        // N_KNOWN and
    C_KEPT_NL,        // N_LABELS when it was captured: the dense node numbering the kept columns are written in
    C_WARM_WINDOWS,   // windows closed on the warm path / by a full rebuild since create (sg_stats)
    C_COLD_WINDOWS,
    // delta windows (round 6): a warm window whose records brought keys the kept set lacks
    C_DELTA_N,        // new edges the warm pass B emitted in this window (zeroed by kc_prepare; consumed when the kept buffers are flipped)
    C_KEPT_BUF,       // which of the two kept-CSR buffers (k_col / k_col2 ...) is current; a full rebuild writes buffer 0, a delta window the other one
    C_DELTA_WINDOWS,  // windows closed warm WITH new edges since create
    C_COUNT = 40
};

// per-workgroup statistics slots written by K1 (one 64-byte line per workgroup: no cross-WG
// contention), reduced at window close.
enum { WS_TMIN = 0, WS_TMAX, WS_MAXLABEL, WS_DROPPED_SRC, WS_DROPPED_CAP, WS_MISROUTED, WS_ACCEPTED, WS_PAD, WS_WORDS };
#define SG_MAX_K1_WGS 2048

// node statistics words (SUM block), see include/servicegraph.h SG_NODE_STAT_SUM_WORDS
enum { ST_OUT_DEG = 0, ST_IN_DEG, ST_OUT_CNT, ST_IN_CNT, ST_OUT_ERR, ST_IN_ERR, ST_OUT_SUM, ST_IN_SUM, ST_OUT_SSQ, ST_IN_SSQ, ST_OUT_ALIVE, ST_IN_ALIVE };

#define SG_MEAN_SLOTS 16
#define SG_ACT_NONE 0xFFFFFFFFull          // ctr[C_ACT_L]: the window has no active node lists (yet)
// One out-degree counter per 32-byte sector: device-scope atomics serialise per sector (~12 ns each,
// profiles/r01_atomic_probe.txt), so neighbouring nodes must not share one.
#define SG_DEG_STRIDE 8
// ... and every row has SG_DEG_REP counters (replica = partition & (REP - 1)), one sector each: the edges of
// a hub row arrive from hundreds of partitions at once and would serialise on one counter (a 3000-edge row:
// ~36 us at the tail of k1b_merge).  k2_rowptr (multi-workgroup) sums the replicas of a row and turns them
// into offsets inside the row.
#define SG_DEG_REP 8
#define SG_DEG_IDX(f, r) (((size_t)(f) * SG_DEG_REP + (r)) * SG_DEG_STRIDE)
#define SG_LB_RESIDENT 256       // workgroups of a look-back kernel that are certainly all resident (one per CU); above it the workgroups order themselves by ticket
#define K2_RP_ROWS 256           // rows per workgroup of k2_rowptr (a multiple of 128; 1024 threads: 8 lanes per row, 2 passes)

// phase stamps for kernel tuning (off unless SG_ABLATE & 0x100): 100 MHz wall clock, thread 0 of a workgroup
#define SG_L1P_TAB 4096u
// Tuning / ablation hooks (SG_ABLATE bits, phase stamps) exist in the DEVELOPMENT build only (-DSG_DEV_KNOBS: alaz_amd/build.py writes it to
// lib/libservicegraph_dev.so, which the tools and the A/B tests of alternative kernel paths load); in the shipped library they are
// compiled out of the kernels and the host reads no SG_* environment variable.
#ifdef SG_DEV_KNOBS
#define SG_ABL(d, bits) (((d).ablate & (bits)) != 0)
#else
#define SG_ABL(d, bits) (false)
#endif
#define SG_STAMP(d, kid, k) do { if (SG_ABL(d, 0x100u) && threadIdx.x == 0 && blockIdx.x < 4096) (d).dbg[((size_t)(kid) * 4096 + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)

// Everything the kernels need, passed by value as one kernel argument.
struct Dev {
    // ---- persistent across windows ----
    const u64* iptab;  u32 ipmask;             // main join table: ipmask + 1 entries = (ipmask + 1) / 2 buckets
    const u64* iptab2; u32 ipmask2;            // pod ids of IPs that are in both maps
    const uint8_t* kind;            // [max_known] SG_NODE_POD / SG_NODE_SERVICE
    const u64* jl1;  u32 jl1mask;              // block table level 1 (global copy; k1a stages it in LDS)
    const u32* jl2;  u32 jl2_words;            // block table level 2: jl2_words = blocks * 256 (0 = no block table)
    u32 ck_n;                                  // IPs in the (residual) cuckoo table iptab; 0 = never probed
    u32 jstage_bytes;                          // bytes k1a copies into LDS: level 1 (jl1mask + 1 entries), then the used blocks of level 2 if they fit
    u32 jl2_in_lds;                            // level 2 is part of that
    u32 max_known, max_labels, max_obip;
    u32 rank, world;
    // ---- open window (K1 state) ----
    u64* ekeys;  u64* eacc;  u32 emask;      // edge table: key, 4 x u64 accumulators per slot
    u64* obkeys; u32 obmask;                  // outbound-ip table: key = ip | 1<<32, 0 = empty
    u64* wgstat;                              // [SG_MAX_K1_WGS][WS_WORDS]
    u64* ctr;                                 // [C_COUNT]
    u64 max_edges;
    // ---- partitioned K1 (variant 0): per-(partition, workgroup) record slabs ----
    u32 variant;                              // 0 = partitioned (LDS aggregation), 1 = global edge table + atomics
    u32 np, nwg;                              // partitions (power of two), pass-A workgroups
    u32 ss, sa;                               // slab piece capacity: single / aggregate records
    u32 pslots;                               // 16-byte slots per piece = ss + 3 * sa
    uint4* slab_s;                            // [np][nwg][pslots]: slots [0, ss) singles {to, from, dur.lo, dur.hi | err<<31 | edge-only<<30};
                                              //   slot ss + 3r: aggregate r {key, cnt|err<<32, sum_ns, max_ns, sumsq_us} (40 of 48 bytes)
    u32*   hdr;                               // [np][nwg] records in piece (p, w): n_single | n_aggregate << 20
    u32 k1a_ct;                               // pass A: LDS edge-cache slots (power of two; bucket = 2 adjacent slots)
    // ---- narrow-record K1 (variant 0 default): 8-byte records, partition by a bijective key mix (sg_hash.h sg_kmix) ----
    u32 narrow;                               // 1 = k1a_tile_partition / k1b_stream_merge run; 0 = the 16-byte-record kernels (k1a_partition / k1b_merge)
    u32 nb, pb, rb;                           // bits per compact endpoint index, log2(np), bits of the in-partition remainder (2 nb - pb <= 31)
    u64*   slab8;                             // [np][nwg][punits] 8-byte units.  Piece (p, w): units [0, sn) narrow records {dur u32, rem | err << 31};
                                              //   then sw wide singles of 2 units {mixed key, dur | err << 63 | edge-only << 62}; then sa aggregates of 5
                                              //   units {mixed key, cnt | err << 32, sum_ns, max_ns, sumsq_us}
    uint2* hdr8;                              // [np][nwg] {narrow records, wide singles | aggregates << 16} in piece (p, w)
    u32 k1b_split;                            // narrow pass B: workgroups (sub-tables) per partition, 1 or 2; output partitions npb = np * k1b_split
    u32 npb;                                  // partitions of the pass-B OUTPUT (e_from / e_to / acc_src / e_rank / part_n): np, or np * k1b_split
    u32 sn, sw, punits;                       // piece geometry (sa is shared with the 16-byte layout); punits = sn + 2 sw + 5 sa, a multiple of 16
    u32 k1b_ht;                               // pass B: LDS table slots per partition (power of two)
    // f-3 (SG_CFG_EDGE_HISTOGRAM): per-edge log2 latency histogram, SG_HIST_BINS u32 bins
    u32 hist;                                 // 0 = off (the kernels' HIST = false instantiations run)
    u32 agg_slots;                            // 16-byte slots per aggregate record: 3, or 5 with the 16 x u16 bins of the launch
    u32* hist_src;                            // bins by slot: [np * pcap][16] (variant 0, written by pass B) or [edge table][16] (variant 1)
    u32* hist_csr;                            // [max_edges][16] bins in CSR (= row) order
    u64*   ovf;  u32 ovf_cap;                 // overflow records [ovf_cap][5] (pieces that ran full)
    u32*   ovf_p;                             // [ovf_cap] partition of each overflow record (pass B filters on 4 bytes, not 40)
    u32*   part_n;                            // [np] distinct edges per partition
    u32 pcap;                                 // edge capacity per partition
    u64*   acc_src;                           // accumulators by slot: eacc (variant 1) or partition output (variant 0)
    u32*   longrows;                          // [ncap] rows with more than 64 edges (work list of the row sort)
    u32*   e_rank;                            // [np*pcap] position of the edge inside its row (arrival order)
    u64* in_part;                             // k3_in_part -> k3_in_reduce: [node ranges][edge slices][K3_IN_NR][6] partial in-statistics
    u32 batch_state;                          // per launch: k1a_partition 1 = first batch of the window (piece headers are
                                              // not read, every workgroup rewrites all of its headers); k1b_merge 2 = the
                                              // window had no batch (pieces hold the previous window: ignore them)
    u32 ablate;                               // SG_ABLATE: 0x100 = record phase stamps (SG_STAMP); 0 in production
    u64* dbg;                                 // phase time stamps (SG_ABLATE & 0x100): [kernel 0..3][4096 workgroups][8]
    u32 k2_sortw;                             // k2_rowsort_gather: words of each of its two LDS arrays (>= K2_SORT_LDS)
    u32 k1a_rot;                              // pass A: the workgroup (unit) that takes this launch's first chunk (tile): successive small batches must not all land on workgroup 0's pieces
    u32* k1a_ticket; u32 k1a_ticket_base;     // k1a_team_partition: the window slot's tile-ticket counter (never reset) and its value before this launch's first ticket
    u64* clk;                                 // [4] shader clock under load: {shader cycles, 100 MHz ticks} summed over pass-A launches (workgroup 0), then the spin probe's pair
    u64* alive_keys; u32 alive_cap;           // edge keys of the window's SG_EV_ALIVE records (marked onto the CSR at close)
    u32* alive_csr;                           // [max_edges] CSR order: open connections per edge
    u32* act_l; u32* act_p;                   // [ncap] world > 1: active node lists (ascending), built with the halo requests
    u64* rp_tot;                              // [ceil((ncap+1)/K2_RP_ROWS)] k2_rowptr: (epoch << 32 | rows' edge total) per workgroup
    // ---- warm windows (variant 0, 8-byte records, no histogram): state carried from one window to the next ----
    // A service map's edge set hardly changes from window to window.  The last COLD window (full rebuild) leaves behind pass B's table
    // image — key per slot — and, per slot, the position its edge got in the CSR; its CSR (row pointers, columns, sources) is kept as
    // the "kept CSR".  A WARM window seeds pass B's LDS tables with the image: every record finds its key where it was, the
    // accumulators go straight to k_acc[kept position], and kw_compact turns the kept CSR minus the untouched edges into the window's
    // CSR in one stable pass (no degree histogram, no row scan over partitions, no scatter, no row sort).  Any key the image lacks, a
    // changed node numbering or a raw outbound IP sends the window down the full rebuild, which captures the state anew.
    u32 warm;                                 // 1 = this engine keeps the state
    u32 kept_compact;                         // this launch builds / merges the KEPT state (set per launch by the host: a warm engine's pass B and rebuild chain, not its
                                              // "plain" closes): node ids are then COMPACT indices — known id | max_known + label — unless the window has raw outbound IPs
                                              // (round 6: the kept CSR no longer depends on N_KNOWN / N_LABELS, so a new pod or Host label does not cost a rebuild)
    u32* wk_keys;                             // [npb][k1b_ht] pass B's table image of the last cold window (all ones = empty)
    u32* wk_pos;                              // [npb][k1b_ht] the slot's edge: partition-output index (cold pass B), then its kept-CSR position (kw_capture); SG_NONE = none
    u32* pos_of_slot;                         // [npb * pcap] CSR position the row sort gave the edge of a partition-output slot (cold windows)
    u64* k_acc;                               // [max_edges][4] accumulators by kept position, rewritten by every warm pass B; bit 63 of word 2 (max_ns < 2^62) = touched in this window
    u32* k_col; u32* k_from; u32* k_rowptr;   // the kept CSR: [max_edges], [max_edges], [ncap + 1]
    u64* kw_tot;                              // [max_edges / KW_CH + 2] kw_compact: (epoch << 32 | touched edges) per chunk
    // Delta windows (round 6).  A key the image lacks no longer sends the window down the full rebuild: the warm pass B inserts it, emits
    // the NEW edges in the cold format (partition outputs e_from / e_to / acc_src / e_rank, part_n = new edges of the partition, row ranks
    // from deg2), the rebuild chain runs on these few edges only and leaves a sorted DELTA CSR (dc_*), and kw_compact merges it into the
    // window's CSR and into the kept CSR — written to the OTHER kept buffer with every position shifted by the new edges before it; the
    // image's positions (wk_pos) follow through k_slot.  The kept set only grows; a full rebuild starts it afresh.
    u32* k_col2; u32* k_from2; u32* k_rowptr2; // the second kept-CSR buffer (ctr[C_KEPT_BUF] says which is current)
    u32* k_slot;                              // [2][npb * pcap] kept position -> image index (partition * k1b_ht + slot), per buffer
    u32* dc_rowptr; u32* dc_col; u32* dc_from; // the delta CSR: [ncap + 2], [npb * pcap], [npb * pcap]
    u64* dc_acc;                              // [npb * pcap][4]
    u32* dc_slot;                             // [npb * pcap] delta position -> partition-output slot (written by the row sort)
    u32* dc_ip;                               // [npb * pcap] delta position -> insertion point in the kept CSR (kw_compact's scratch)
    u32* dl_img;                              // [npb * pcap] partition-output slot of a new edge -> its image index
    u32* deg2;                                // [ncap + 1][SG_DEG_REP] row-degree replicas of the new edges (deg belongs to a full rebuild; zeroed by the window reset)
    u32* k1b_cnt;                             // [np] narrow records pass B read per partition in the LAST window (kc_prepare consumes and zeroes them)
    u32* k1b_order;                           // [np] the partitions by those counts, largest first: pass B's workgroup i takes partition k1b_order[i] (kc_prepare, every window)
    u64* host_note;                           // page-locked HOST memory (mapped): [0] = sequence number of the last window kw_compact closed, [1] = its C_COLD
                                              // and C_N_OBIP << 8 — how the host learns, without ever waiting for the device, which path its windows take
    u32* lb_ticket;                           // [4] self-resetting workgroup tickets of the look-back kernels whose grid exceeds SG_LB_RESIDENT ([0] k2_rowptr, [1] kw_compact)
    u64* k6_tot;                              // [ceil(ncap/1024) + 1][16] k6_halo_lists: (epoch << 32 | members) per workgroup and list
    // ---- closed window ----
    u32* ob_sorted;                           // [max_obip] ascending distinct raw IPs
    u32* tile_cnt;  u32* tile_off;            // compaction scratch
    u32* e_slot;    u32* e_from;  u32* e_to;  // [max_edges] compacted (table order)
    u32* deg;       u32* rowptr;  u32* cursor;// [ncap+1]
    // Row degrees and in-row positions WITHOUT device atomics (variant 0, node spaces whose u32 counters fit one workgroup's LDS):
    // k2_deg_hist — workgroup g of dh_g counts the sources of dh_ppw consecutive output partitions in LDS (the returning LDS add is the
    // edge's rank among the (g, source) edges -> e_rank) and writes its counts to dh_hist[g][.]; k2_rowptr turns every row's column of
    // counts into offsets inside the row; the scatter adds rowptr + offset(g, source) + rank.  dh_g = 0: pass B takes the rank with
    // one returning device atomic per edge on deg[source][replica] (1 M of them at C3: 23-38 us of pass B).
    u32 dh_g, dh_ppw, dh_ns;                  // histogram workgroups (0 = off), output partitions per workgroup, row stride of dh_hist (>= ncap + 1, a multiple of 64)
    u32* dh_hist;                             // [dh_g][dh_ns]
    u32* col;                                 // [max_edges] CSR order: destination (written by the row sort, sorted inside every row)
    uint2* cs;                                // [max_edges] {destination, table slot} in row order, unsorted: ONE 8-byte scattered write per edge (scatter -> row sort)
    u32* csr_from;                            // [max_edges] CSR order: source (row id per edge)
    u32* sort_k;    u32* sort_v;              // [2*max_edges] scratch for rows longer than the LDS sort
    u64* acc_csr;                             // [max_edges][4]
    u64* st_sum;    u64* st_max;              // [ncap][SG_NODE_STAT_SUM_WORDS], [ncap][2]
    float* x0;                                // [ncap][32]
    float* h[SG_MAX_LAYERS + 1];              // h[l] = output of layer l (l>=1): [ncap][64]
    float* P; float* Q;                       // [ncap][64]
    float* nmean;                             // [ncap][64] neighbour means of the layer being computed (k4_gather -> k4_sage_layer)
    uint2* hub_items; u32 hub_cap;            // [hub_cap] {row, block}: the 512-neighbour blocks of the rows longer than one block, one work item each
    u32* hub_base;                            // [ncap] first work item of a hub row
    float* hub_part;                          // [hub_cap][64] block sums of the layer being computed (k4_gather -> k4_sage_layer adds them in block order)
    double* row_mu; double* row_sd;           // [ncap] mean / std (us) of a source's out-events: written by the row sort, read once per edge by edge_features
    const float* l1p_tab;                     // [SG_L1P_TAB] (float)log1p((double)i), filled on the device at create: small counts skip the fp64 log1p
    float* efeat;                             // [max_edges][8]
    float* latz; float* errr;                 // [max_edges]
    sg_edge_out* rows;                        // [max_edges]
    const float* W;                           // weights blob
    u32 ncap; u32 layers;
};
