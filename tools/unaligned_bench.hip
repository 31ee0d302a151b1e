// Bandwidth of 8-byte / 4-byte / 2-byte global loads and stores per lane at aligned and 2-byte-misaligned addresses (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/unaligned_bench.hip -o /tmp/unaligned_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

template <int BYTES>
__global__ void __launch_bounds__(256) rd(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, long n_items, int mis, int do_store) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n_items; i += stride) {
        const uint8_t *p = src + mis + i * BYTES;
        if (BYTES == 8) { unsigned long long v; __builtin_memcpy(&v, __builtin_assume_aligned(p, 2), 8); acc += v; if (do_store) __builtin_memcpy(__builtin_assume_aligned(dst + mis + i * 8, 2), &v, 8); }
        if (BYTES == 4) { unsigned v; __builtin_memcpy(&v, p, 4); acc += v; if (do_store) __builtin_memcpy(dst + mis + i * 4, &v, 4); }
        if (BYTES == 2) { unsigned short v; __builtin_memcpy(&v, __builtin_assume_aligned(p, 2), 2); acc += v; if (do_store) __builtin_memcpy(__builtin_assume_aligned(dst + mis + i * 2, 2), &v, 2); }
        if (BYTES == 1) { uint8_t v = *p; acc += v; if (do_store) dst[mis + i] = v; }
    }
    if (acc == 0x123456789abcdefull) dst[0] = 1;
}

int main() {
    const long N = 1l << 31;   // 2 GiB
    uint8_t *a, *b;
    hipMalloc(&a, N + 64); hipMalloc(&b, N + 64);
    hipMemset(a, 1, N + 64); hipMemset(b, 0, N + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int store = 0; store <= 1; store++)
    for (int bytes : {1, 2, 4, 8})
    for (int mis : {0, 2, 1}) {
        if ((bytes == 2 || bytes == 8) && mis == 1) continue;
        if (bytes == 1 && mis) continue;
        const long items = N / bytes;
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (bytes == 1) rd<1><<<8192, 256>>>(a, b, items, mis, store);
            if (bytes == 2) rd<2><<<8192, 256>>>(a, b, items, mis, store);
            if (bytes == 4) rd<4><<<8192, 256>>>(a, b, items, mis, store);
            if (bytes == 8) rd<8><<<8192, 256>>>(a, b, items, mis, store);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s %d B/lane, misaligned by %d: %.3f ms  %.0f GB/s\n", store ? "copy" : "read", bytes, mis, best, (store ? 2.0 : 1.0) * N / best / 1e6);
    }
    return 0;
}
