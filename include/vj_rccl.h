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
 * Every rank ends up with the same list, sorted by (frame, scale_idx, y, x).  Two collectives: the counts, then one
 * padded buffer per rank (payloads are a few KB: latency-, not bandwidth-bound).
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

/* Collective over `comm` (call it from every rank, inside ncclGroupStart/End when one thread drives several ranks is NOT
 * supported: it synchronises `stream` between its two collectives).  *all is malloc'ed; the caller free()s it.
 * Returns VJ_OK, VJ_ERR_NOMEM or VJ_ERR_HIP (a HIP / RCCL call failed).                                              */
static inline int vj_rccl_allgather_rects(ncclComm_t comm, hipStream_t stream, const vj_rect* local, uint32_t n_local,
                                          int n_ranks, vj_rect** all, uint32_t* n_all) {
    int rc = VJ_ERR_HIP;
    unsigned long long* d_counts = NULL;   /* [n_ranks + 1]: gathered counts, then this rank's own */
    char* d_buf = NULL;
    vj_rect* h_all = NULL;
    unsigned long long* h_counts = (unsigned long long*)malloc(sizeof(unsigned long long) * (size_t)(n_ranks + 1));
    *all = NULL;
    *n_all = 0;
    if (!h_counts) return VJ_ERR_NOMEM;
    if (hipMalloc((void**)&d_counts, sizeof(unsigned long long) * (size_t)(n_ranks + 1)) != hipSuccess) goto done;
    h_counts[n_ranks] = n_local;
    if (hipMemcpyAsync(d_counts + n_ranks, h_counts + n_ranks, sizeof(unsigned long long), hipMemcpyHostToDevice, stream) != hipSuccess) goto done;
    if (ncclAllGather(d_counts + n_ranks, d_counts, 1, ncclUint64, comm, stream) != ncclSuccess) goto done;
    if (hipMemcpyAsync(h_counts, d_counts, sizeof(unsigned long long) * (size_t)n_ranks, hipMemcpyDeviceToHost, stream) != hipSuccess) goto done;
    if (hipStreamSynchronize(stream) != hipSuccess) goto done;
    {
        unsigned long long cap = 1, total = 0;
        for (int r = 0; r < n_ranks; ++r) {
            if (h_counts[r] > cap) cap = h_counts[r];
            total += h_counts[r];
        }
        const size_t slot = (size_t)cap * sizeof(vj_rect);
        /* [n_ranks slots: receive] [1 slot: this rank's padded send buffer] */
        if (hipMalloc((void**)&d_buf, slot * (size_t)(n_ranks + 1)) != hipSuccess) goto done;
        if (n_local && hipMemcpyAsync(d_buf + slot * (size_t)n_ranks, local, (size_t)n_local * sizeof(vj_rect), hipMemcpyHostToDevice, stream) != hipSuccess) goto done;
        if (ncclAllGather(d_buf + slot * (size_t)n_ranks, d_buf, slot, ncclChar, comm, stream) != ncclSuccess) goto done;
        h_all = (vj_rect*)malloc(total ? (size_t)total * sizeof(vj_rect) : 1);
        if (!h_all) { rc = VJ_ERR_NOMEM; goto done; }
        size_t at = 0;
        for (int r = 0; r < n_ranks; ++r) {
            if (h_counts[r] && hipMemcpyAsync(h_all + at, d_buf + slot * (size_t)r, (size_t)h_counts[r] * sizeof(vj_rect), hipMemcpyDeviceToHost, stream) != hipSuccess) goto done;
            at += (size_t)h_counts[r];
        }
        if (hipStreamSynchronize(stream) != hipSuccess) goto done;
        qsort(h_all, (size_t)total, sizeof(vj_rect), vj_rccl_rect_cmp_);
        *all = h_all;
        *n_all = (uint32_t)total;
        h_all = NULL;
        rc = VJ_OK;
    }
done:
    free(h_all);
    free(h_counts);
    if (d_counts) (void)hipFree(d_counts);
    if (d_buf) (void)hipFree(d_buf);
    return rc;
}

#ifdef __cplusplus
}
#endif
#endif /* VJ_RCCL_H_ */
