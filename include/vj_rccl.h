/*
 * vj_rccl.h — the one collective of the multi-GPU detect path, for C / C++ hosts: an all-gather of every rank's
 * detection rectangles with RCCL (ncclAllGather over xGMI).  Header-only on purpose: libvjhip.so does not link
 * librccl; the host application that includes this header does (-lrccl).  The reference has no multi-device code at all
 * (clodInitEnvironment takes a device_index, clod.cpp:72-100, and that is it); SURVEY.md §8e defines this step.
 *
 * Usage, one rank per GPU (processes with ncclCommInitRank, or one process with ncclCommInitAll and a thread per device):
 *     vj_shard_frames(n_frames, n_ranks, rank, &first, &count);            // or vj_shard_scales for one large frame
 *     vj_detect(env, cascade, frames + first, count, &params, &result);    // rect.frame is local: add `first`
 *     vj_rccl_allgather_rects(comm, stream, result.rects, result.count, n_ranks, &all, &n_all);
 * Every rank ends up with the same list, sorted by (frame, scale_idx, y, x).  ONE collective per step: fixed-capacity
 * `[count | rects x cap]` blocks, regrown on overflow (vj_rccl_gatherer; payloads are a few KB: latency-, not
 * bandwidth-bound).  A loop over steps keeps one vj_rccl_gatherer; vj_rccl_allgather_rects is the one-shot form.
 */
#ifndef VJ_RCCL_H_
#define VJ_RCCL_H_

#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "vj.h"

#ifdef __cplusplus
extern "C" {
#endif

static int vj_rccl_rect_cmp_(const void* a_, const void* b_) {
    const vj_rect* a = (const vj_rect*)a_;
    const vj_rect* b = (const vj_rect*)b_;
    if (a->frame != b->frame) return a->frame < b->frame ? -1 : 1;
    if (a->scale_idx != b->scale_idx) return a->scale_idx < b->scale_idx ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    return 0;
}

/* The gatherer keeps its buffers and its capacity across steps: every rank contributes ONE fixed-capacity block
 * `[uint32 count | padding to 32 bytes | vj_rect x cap]` and one ncclAllGather moves all of them.  When a
 * rank's count exceeds the capacity every rank reads that in the gathered headers, all double the capacity the same way and
 * repeat the step's collective: a steady workload costs one collective, one copy in and one copy out per step.
 * `n_ranks` is what the COMMUNICATOR reports (ncclCommCount), not what the caller believes.                            */
typedef struct vj_rccl_gatherer {
    ncclComm_t comm;
    hipStream_t stream;
    int n_ranks;               /* ncclCommCount(comm) */
    uint32_t cap;              /* rectangles per rank and step before the buffers grow */
    char* d_send;              /* one block */
    char* d_recv;              /* n_ranks blocks */
    char* h_recv;              /* page-locked mirror of d_recv */
    char* h_send;              /* page-locked staging of this rank's block */
    uint64_t n_collectives;    /* ncclAllGather calls so far */
    float last_ms;             /* device time of the last step's copies + collective (HIP events on `stream`) */
    hipEvent_t ev0, ev1;
} vj_rccl_gatherer;

#define VJ_RCCL_HEADER_BYTES 32u
static inline size_t vj_rccl_block_bytes_(uint32_t cap) { return VJ_RCCL_HEADER_BYTES + (size_t)cap * sizeof(vj_rect); }

static inline void vj_rccl_gatherer_free_buffers_(vj_rccl_gatherer* g) {
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    if (g->h_recv) (void)hipHostFree(g->h_recv);
    if (g->h_send) (void)hipHostFree(g->h_send);
    g->d_send = g->d_recv = g->h_recv = g->h_send = NULL;
}

static inline int vj_rccl_gatherer_alloc_(vj_rccl_gatherer* g, uint32_t cap) {
    vj_rccl_gatherer_free_buffers_(g);
    const size_t blk = vj_rccl_block_bytes_(cap);
    if (hipMalloc((void**)&g->d_send, blk) != hipSuccess) return VJ_ERR_NOMEM;
    if (hipMalloc((void**)&g->d_recv, blk * (size_t)g->n_ranks) != hipSuccess) return VJ_ERR_NOMEM;
    if (hipHostMalloc((void**)&g->h_recv, blk * (size_t)g->n_ranks, hipHostMallocDefault) != hipSuccess) return VJ_ERR_NOMEM;
    if (hipHostMalloc((void**)&g->h_send, blk, hipHostMallocDefault) != hipSuccess) return VJ_ERR_NOMEM;
    g->cap = cap;
    return VJ_OK;
}

static inline void vj_rccl_gatherer_destroy(vj_rccl_gatherer* g) {
    if (!g) return;
    vj_rccl_gatherer_free_buffers_(g);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    memset(g, 0, sizeof(*g));
}

/* `cap0`: rectangles per rank the first buffers hold (0: 1024).  The device of `stream` must be current. */
static inline int vj_rccl_gatherer_init(vj_rccl_gatherer* g, ncclComm_t comm, hipStream_t stream, uint32_t cap0) {
    memset(g, 0, sizeof(*g));
    g->comm = comm;
    g->stream = stream;
    if (ncclCommCount(comm, &g->n_ranks) != ncclSuccess || g->n_ranks < 1) return VJ_ERR_HIP;
    if (hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) return VJ_ERR_HIP;
    const int rc = vj_rccl_gatherer_alloc_(g, cap0 ? cap0 : 1024u);
    if (rc) vj_rccl_gatherer_destroy(g);
    return rc;
}

/* One step: collective over g->comm (every rank calls it).  *all is malloc'ed, sorted by (frame, scale_idx, y, x); the
 * caller free()s it.  Returns VJ_OK, VJ_ERR_NOMEM or VJ_ERR_HIP (a HIP / RCCL call failed).                               */
static inline int vj_rccl_gatherer_run(vj_rccl_gatherer* g, const vj_rect* local, uint32_t n_local, vj_rect** all, uint32_t* n_all) {
    *all = NULL;
    *n_all = 0;
    for (int attempt = 0; attempt < 32; ++attempt) {
        const size_t blk = vj_rccl_block_bytes_(g->cap);
        const uint32_t k = n_local < g->cap ? n_local : g->cap;
        memset(g->h_send, 0, VJ_RCCL_HEADER_BYTES);
        memcpy(g->h_send, &n_local, sizeof(uint32_t));
        if (k) memcpy(g->h_send + VJ_RCCL_HEADER_BYTES, local, (size_t)k * sizeof(vj_rect));
        if (hipEventRecord(g->ev0, g->stream) != hipSuccess) return VJ_ERR_HIP;
        if (hipMemcpyAsync(g->d_send, g->h_send, VJ_RCCL_HEADER_BYTES + (size_t)k * sizeof(vj_rect), hipMemcpyHostToDevice, g->stream) != hipSuccess) return VJ_ERR_HIP;
        if (ncclAllGather(g->d_send, g->d_recv, blk, ncclChar, g->comm, g->stream) != ncclSuccess) return VJ_ERR_HIP;
        ++g->n_collectives;
        if (hipMemcpyAsync(g->h_recv, g->d_recv, blk * (size_t)g->n_ranks, hipMemcpyDeviceToHost, g->stream) != hipSuccess) return VJ_ERR_HIP;
        if (hipEventRecord(g->ev1, g->stream) != hipSuccess) return VJ_ERR_HIP;
        if (hipStreamSynchronize(g->stream) != hipSuccess) return VJ_ERR_HIP;
        (void)hipEventElapsedTime(&g->last_ms, g->ev0, g->ev1);
        uint32_t mx = 0;
        uint64_t total = 0;
        for (int r = 0; r < g->n_ranks; ++r) {
            uint32_t c;
            memcpy(&c, g->h_recv + blk * (size_t)r, sizeof(uint32_t));
            if (c > mx) mx = c;
            total += c;
        }
        if (mx > g->cap) {      /* every rank read the same headers: the same new capacity everywhere, then once more */
            uint32_t cap = g->cap;
            while (cap < mx) cap *= 2u;
            const int rc = vj_rccl_gatherer_alloc_(g, cap);
            if (rc) return rc;
            continue;
        }
        vj_rect* h_all = (vj_rect*)malloc(total ? (size_t)total * sizeof(vj_rect) : 1);
        if (!h_all) return VJ_ERR_NOMEM;
        size_t at = 0;
        for (int r = 0; r < g->n_ranks; ++r) {
            uint32_t c;
            memcpy(&c, g->h_recv + blk * (size_t)r, sizeof(uint32_t));
            if (c) memcpy(h_all + at, g->h_recv + blk * (size_t)r + VJ_RCCL_HEADER_BYTES, (size_t)c * sizeof(vj_rect));
            at += c;
        }
        qsort(h_all, (size_t)total, sizeof(vj_rect), vj_rccl_rect_cmp_);
        *all = h_all;
        *n_all = (uint32_t)total;
        return VJ_OK;
    }
    return VJ_ERR_LIMIT;
}

/* One-shot convenience: a gatherer for a single step.  `n_ranks` must be what the communicator reports. */
static inline int vj_rccl_allgather_rects(ncclComm_t comm, hipStream_t stream, const vj_rect* local, uint32_t n_local,
                                          int n_ranks, vj_rect** all, uint32_t* n_all) {
    vj_rccl_gatherer g;
    int rc = vj_rccl_gatherer_init(&g, comm, stream, n_local > 1024u ? n_local : 1024u);
    if (rc) return rc;
    if (g.n_ranks != n_ranks) {
        vj_rccl_gatherer_destroy(&g);
        return VJ_ERR_ARG;
    }
    rc = vj_rccl_gatherer_run(&g, local, n_local, all, n_all);
    vj_rccl_gatherer_destroy(&g);
    return rc;
}

#ifdef __cplusplus
}
#endif
#endif /* VJ_RCCL_H_ */
