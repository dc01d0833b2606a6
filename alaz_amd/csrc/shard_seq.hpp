// shard_seq.hpp — the per-window sequence of a SHARDED ServiceGraph engine, as one function.
//
// One process per GPU; the graph is hash-sharded by source node, so K1 needs no exchange.  Closing a window
// takes a fixed sequence of local stages and fixed-size collectives — nothing in it waits for the device, the host only
// enqueues:
//
//   obip_list -> all_gather(raw outbound IPs)  -> close (identical node numbering on every shard)
//             -> all_reduce SUM / MAX of the integer node statistics (exact, order-free; one grouped launch) -> features
//             -> halo_build -> all_to_all(request lists)
//             -> for every layer: layer -> pack -> all_to_all(requested rows: copied, never reduced) -> unpack
//             -> score (+ window reset)
//
// The function is shared, header-only, by the two libraries: libservicegraph.so runs it with the HIP stages and an RCCL
// communicator (sg_window_run_sharded: ONE C call per window instead of ~15 ctypes calls and six torch.distributed calls from
// Python), libsgdatastore.so exports it with caller-supplied stage / communicator callbacks (sgh_run_sharded_window), which
// is how the 2-process gloo test drives exactly this sequence on CPU.  No HIP, no torch in this file.
#pragma once
#include <stddef.h>
#include <stdint.h>

extern "C" {
// Collectives over `world` ranks; every call enqueues on the communicator's stream and returns (0 = ok).
typedef struct sg_shard_comm {
    void* ctx;
    int (*all_gather)(void* ctx, const void* send, void* recv, size_t bytes_per_rank);          // recv = [world][bytes_per_rank]
    int (*all_reduce_u64)(void* ctx, void* buf, size_t count, int op);                           // in place; op 0 = sum, 1 = max
    int (*all_to_all)(void* ctx, const void* send, void* recv, size_t bytes_per_rank);          // both [world][bytes_per_rank]
    // optional (may be null): the collectives issued between the two calls may be launched as one (RCCL: ncclGroupStart / ncclGroupEnd —
    // the SUM and the MAX all-reduce of the node statistics leave as ONE launch instead of two)
    int (*group_begin)(void* ctx);
    int (*group_end)(void* ctx);
} sg_shard_comm;

// The local stages of one shard and the exchange buffers they fill / read (device memory for the HIP engine).
typedef struct sg_shard_stages {
    void* ctx;
    uint32_t layers, world;
    int (*obip_list)(void* ctx);                  // fills ob_local = [count, ip, ip, ...]
    int (*close_gathered)(void* ctx);             // reads ob_all
    int (*features)(void* ctx);                   // reads the reduced statistics
    int (*halo_build)(void* ctx);                 // fills req = what this shard needs from each owner
    int (*layer)(void* ctx, uint32_t l);
    int (*pack)(void* ctx, uint32_t l);           // rows of layer l for the lists in serve -> rows_out
    int (*unpack)(void* ctx, uint32_t l);         // rows_in -> the feature buffer of layer l, at the ids in req
    int (*score)(void* ctx);
    void* ob_local;  void* ob_all;   size_t ob_bytes;          // per rank
    void* stats_sum; size_t stats_sum_words; void* stats_max; size_t stats_max_words;
    void* req;       void* serve;    size_t list_bytes;        // per rank
    void* rows_out;  void* rows_in;  size_t rows_bytes;        // per rank
} sg_shard_stages;
}

static inline int sg_run_sharded_window(const sg_shard_stages* s, const sg_shard_comm* c) {
    if (!s || !c || !c->all_gather || !c->all_reduce_u64 || !c->all_to_all) return -22;
#define SG_SEQ(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)
    SG_SEQ(s->obip_list(s->ctx));
    SG_SEQ(c->all_gather(c->ctx, s->ob_local, s->ob_all, s->ob_bytes));
    SG_SEQ(s->close_gathered(s->ctx));
    {   // the SUM and the MAX all-reduce as one grouped launch; a failure inside the group still closes it (an open group would swallow
        // every later collective of this communicator)
        const bool grp = c->group_begin && c->group_end;
        if (grp) SG_SEQ(c->group_begin(c->ctx));
        int rc_ = c->all_reduce_u64(c->ctx, s->stats_sum, s->stats_sum_words, 0);
        if (!rc_) rc_ = c->all_reduce_u64(c->ctx, s->stats_max, s->stats_max_words, 1);
        if (grp) { const int re_ = c->group_end(c->ctx); if (!rc_) rc_ = re_; }
        if (rc_) return rc_;
    }
    SG_SEQ(s->features(s->ctx));
    SG_SEQ(s->halo_build(s->ctx));
    SG_SEQ(c->all_to_all(c->ctx, s->req, s->serve, s->list_bytes));
    for (uint32_t l = 0; l < s->layers; l++) {
        SG_SEQ(s->layer(s->ctx, l));
        SG_SEQ(s->pack(s->ctx, l + 1));
        SG_SEQ(c->all_to_all(c->ctx, s->rows_out, s->rows_in, s->rows_bytes));
        SG_SEQ(s->unpack(s->ctx, l + 1));
    }
    SG_SEQ(s->score(s->ctx));
#undef SG_SEQ
    return 0;
}
