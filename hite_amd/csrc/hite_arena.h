// hite_arena.h -- grow-only device arenas (no hipMalloc on the steady-state path), shared by the fine-stage
// pipeline and the copy finder.
#pragma once
#include "hite_common.h"
#include <vector>

// ---------------------------------------------------------------------------------------------
// arenas (grow-only, reset per call; no hipMalloc on the steady-state path)
// ---------------------------------------------------------------------------------------------
struct Arena {
    std::vector<void *> chunks;
    std::vector<size_t> caps;
    size_t cur = 0;   // chunk being filled
    size_t off = 0;   // bytes used in it
};

static int arena_alloc(hite_ctx *ctx, Arena &a, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    // first chunk (from the current one on) that still has room
    while (a.cur < a.chunks.size() && a.off + bytes > a.caps[a.cur]) { a.cur++; a.off = 0; }
    if (a.cur >= a.chunks.size()) {
        // geometric growth: a new chunk is at least as large as everything allocated so far (few chunks to consolidate)
        size_t total = 0;
        for (size_t c : a.caps) total += c;
        size_t want = bytes > ((size_t)256 << 20) ? bytes : ((size_t)256 << 20);
        if (want < total) want = total;
        void *p = nullptr;
        HITE_CHECK(ctx, hipMalloc(&p, want));
        a.chunks.push_back(p);
        a.caps.push_back(want);
        a.cur = a.chunks.size() - 1;
        a.off = 0;
    }
    *out = (uint8_t *)a.chunks[a.cur] + a.off;
    a.off += bytes;
    return HITE_OK;
}
// soft reset: hand the same memory out again (stream order protects the reuse).
// hard reset (start of a call): if the arena grew into several chunks, replace them by one chunk of
// the total capacity, so that the steady state never calls hipMalloc.
static int arena_reset(hite_ctx *ctx, Arena &a, bool hard) {
    if (hard && a.chunks.size() > 1) {
        HITE_CHECK(ctx, hipDeviceSynchronize());
        size_t total = 0;
        for (size_t c : a.caps) total += c;
        for (void *p : a.chunks) (void)hipFree(p);
        a.chunks.clear();
        a.caps.clear();
        void *p = nullptr;
        HITE_CHECK(ctx, hipMalloc(&p, total));
        a.chunks.push_back(p);
        a.caps.push_back(total);
    }
    a.cur = 0;
    a.off = 0;
    return HITE_OK;
}
static void arena_free(Arena &a) {
    for (void *p : a.chunks) (void)hipFree(p);
    a.chunks.clear();
    a.caps.clear();
    a.cur = a.off = 0;
}

