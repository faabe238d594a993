// Micro-benchmarks, part 2: the "append a layer" pattern with 4 columns per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned char  u8x4  __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned char  u8x8  __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned char  u8x16  __attribute__((ext_vector_type(16)));
typedef unsigned short u16x16 __attribute__((ext_vector_type(16)));

// V columns per thread, U groups per thread.  mode 0: append (R len, mat[l-1], z1[l-1]; W mat[l], z0[l], z1[l], len)
// mode 1: extend (R len, mat[l-1], z1[l-1]; W z1[l-1]).  mode 2: append without the dependent read of plane l-1 (plane index from a constant)
template <typename V8, typename V16, int VW, int U, int MODE>
__global__ __launch_bounds__(256) void k_pat(uint8_t* __restrict__ len, uint8_t* __restrict__ mat, uint16_t* __restrict__ z0, uint16_t* __restrict__ z1,
                                             size_t n, size_t P, int lconst) {
    const size_t ng = n / VW;
    const size_t tile = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * tile + threadIdx.x; base < ng; base += (size_t)gridDim.x * tile) {
        V8 ln[U], tm[U]; V16 tz[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t gi = base + (size_t)u * 256; if (gi < ng) ln[u] = *(const V8*)(len + gi * VW); }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t gi = base + (size_t)u * 256; if (gi >= ng) continue;
            int l = MODE == 2 ? lconst : ln[u][0];
            size_t top = (size_t)(l - 1) * P + gi * VW;
            tm[u] = *(const V8*)(mat + top); tz[u] = *(const V16*)(z1 + top);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t gi = base + (size_t)u * 256; if (gi >= ng) continue;
            int l = MODE == 2 ? lconst : ln[u][0];
            V16 q = tz[u] + (unsigned short)100;
            if (MODE == 1) { *(V16*)(z1 + (size_t)(l - 1) * P + gi * VW) = q; }
            else {
                size_t b = (size_t)l * P + gi * VW;
                *(V8*)(mat + b) = tm[u] + (unsigned char)1; *(V16*)(z0 + b) = tz[u]; *(V16*)(z1 + b) = q;
                *(V8*)(len + gi * VW) = ln[u];   // value unchanged so the benchmark can be repeated
            }
        }
    }
}

int main() {
    const size_t n = (size_t)16384 * 16384, P = n;
    const int L = 8;
    uint8_t *len, *mat; uint16_t *z0, *z1;
    CHK(hipMalloc(&len, n)); CHK(hipMalloc(&mat, P * L)); CHK(hipMalloc(&z0, P * L * 2)); CHK(hipMalloc(&z1, P * L * 2));
    CHK(hipMemset(len, 3, n)); CHK(hipMemset(mat, 4, P * L)); CHK(hipMemset(z0, 1, P * L * 2)); CHK(hipMemset(z1, 1, P * L * 2));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto time = [&](auto launch, const char* name, double bytes) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-52s %8.3f ms  %8.1f GB/s\n", name, best, bytes / best / 1e6);
    };
#define RUN(V8, V16, VW, U, MODE, G, NAME, BYTES) time([&] { hipLaunchKernelGGL((k_pat<V8, V16, VW, U, MODE>), dim3(G), dim3(256), 0, 0, len, mat, z0, z1, n, P, 3); }, NAME, BYTES * n)
    RUN(u8x4, u16x4, 4, 1, 0, 4096, "append V=4 U=1 (10 B/col)", 10.0);
    RUN(u8x4, u16x4, 4, 2, 0, 4096, "append V=4 U=2", 10.0);
    RUN(u8x4, u16x4, 4, 4, 0, 4096, "append V=4 U=4", 10.0);
    RUN(u8x4, u16x4, 4, 4, 0, 16384, "append V=4 U=4 grid 16384", 10.0);
    RUN(u8x4, u16x4, 4, 4, 0, 1024, "append V=4 U=4 grid 1024", 10.0);
    RUN(u8x4, u16x4, 4, 8, 0, 4096, "append V=4 U=8", 10.0);
    RUN(u8x8, u16x8, 8, 1, 0, 4096, "append V=8 U=1", 10.0);
    RUN(u8x8, u16x8, 8, 2, 0, 4096, "append V=8 U=2", 10.0);
    RUN(u8x8, u16x8, 8, 4, 0, 4096, "append V=8 U=4", 10.0);
    RUN(u8x16, u16x16, 16, 1, 0, 4096, "append V=16 U=1", 10.0);
    RUN(u8x16, u16x16, 16, 2, 0, 4096, "append V=16 U=2", 10.0);
    RUN(u8x4, u16x4, 4, 4, 2, 4096, "append V=4 U=4 no dependent plane index", 10.0);
    RUN(u8x8, u16x8, 8, 2, 2, 4096, "append V=8 U=2 no dependent plane index", 10.0);
    RUN(u8x4, u16x4, 4, 4, 1, 4096, "extend V=4 U=4 (6 B/col)", 6.0);
    RUN(u8x8, u16x8, 8, 2, 1, 4096, "extend V=8 U=2", 6.0);
    RUN(u8x16, u16x16, 16, 1, 1, 4096, "extend V=16 U=1", 6.0);
    return 0;
}
