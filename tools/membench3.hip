// Micro-benchmark 3: the access mix of an "append" in the split column form, without the arithmetic:
// per 4 columns R len (4 B) + top (16 B); W plane (16 B) + top (16 B) + topz0 (8 B) + len (4 B)
// (the access mix of an append in an earlier layout that kept the top's bottom and len as streams).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// WMASK bits: 1 plane, 2 top, 4 topz0, 8 len.  INPLACE: top written where it was read.
template <int U, int WMASK, bool INPLACE>
__global__ __launch_bounds__(256) void k_mix(uint8_t* __restrict__ len, uint32_t* __restrict__ top, uint32_t* __restrict__ top2, uint16_t* __restrict__ z0,
                                             uint32_t* __restrict__ plane, size_t n) {
    const size_t ng = n / 4, tile = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * tile + threadIdx.x; base < ng; base += (size_t)gridDim.x * tile) {
        u8x4 ln[U]; u32x4 tv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t gi = base + (size_t)u * 256; if (gi < ng) { ln[u] = *(const u8x4*)(len + gi * 4); tv[u] = *(const u32x4*)(top + gi * 4); } }
#pragma unroll
        for (int u = 0; u < U; u++) {
            size_t gi = base + (size_t)u * 256; if (gi >= ng) continue;
            u32x4 nv = tv[u] + 100u;
            if (WMASK & 1) *(u32x4*)(plane + gi * 4) = tv[u];
            if (WMASK & 2) *(u32x4*)((INPLACE ? top : top2) + gi * 4) = nv;
            if (WMASK & 4) { u16x4 z; z[0] = (unsigned short)tv[u][0]; z[1] = (unsigned short)tv[u][1]; z[2] = (unsigned short)tv[u][2]; z[3] = (unsigned short)tv[u][3]; *(u16x4*)(z0 + gi * 4) = z; }
            if (WMASK & 8) *(u8x4*)(len + gi * 4) = ln[u];
        }
    }
}

// scattered append: the plane a group's old tops go to alternates every `run` groups (run*4 columns)
template <int NPL>
__global__ __launch_bounds__(256) void k_scatter(uint8_t* __restrict__ len, uint32_t* __restrict__ top, uint16_t* __restrict__ z0,
                                                 uint32_t* __restrict__ planes, size_t n, int run, int halfgroup) {
    const size_t ng = n / 4;
    size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ng) return;
    u8x4 ln = *(const u8x4*)(len + gi * 4); u32x4 tv = *(const u32x4*)(top + gi * 4);
    const int p = (int)((gi / run) % NPL);
    if (!halfgroup) *(u32x4*)(planes + (size_t)p * n + gi * 4) = tv;
    else {   // the 4 columns of a group go to two different planes (2 + 2)
        const int p2 = (p + 1) % NPL;
        planes[(size_t)p * n + gi * 4] = tv[0]; planes[(size_t)p * n + gi * 4 + 1] = tv[1];
        planes[(size_t)p2 * n + gi * 4 + 2] = tv[2]; planes[(size_t)p2 * n + gi * 4 + 3] = tv[3];
    }
    *(u32x4*)(top + gi * 4) = tv + 100u;
    u16x4 z; z[0] = (unsigned short)tv[0]; z[1] = (unsigned short)tv[1]; z[2] = (unsigned short)tv[2]; z[3] = (unsigned short)tv[3];
    *(u16x4*)(z0 + gi * 4) = z;
    *(u8x4*)(len + gi * 4) = ln;
}

int main() {
    const size_t n = (size_t)16384 * 16384;
    uint8_t* len; uint32_t *top, *top2, *plane; uint16_t* z0;
    CHK(hipMalloc(&len, n)); CHK(hipMalloc(&top, 4 * n)); CHK(hipMalloc(&top2, 4 * n)); CHK(hipMalloc(&plane, 4 * n)); CHK(hipMalloc(&z0, 2 * n));
    CHK(hipMemset(len, 3, n)); CHK(hipMemset(top, 1, 4 * n)); CHK(hipMemset(top2, 1, 4 * n)); CHK(hipMemset(plane, 1, 4 * n)); CHK(hipMemset(z0, 1, 2 * n));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto time = [&](auto launch, const char* name, double bytes) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-56s %8.3f ms  %8.1f GB/s\n", name, best, bytes * n / best / 1e6);
    };
#define RUN(U, W, IP, G, NAME, BYTES) time([&] { hipLaunchKernelGGL((k_mix<U, W, IP>), dim3(G), dim3(256), 0, 0, len, top, top2, z0, plane, n); }, NAME, BYTES)
    RUN(2, 15, true, 4096, "append mix, in place, U=2, 4096 blocks (16 B/col)", 16.0);
    RUN(2, 15, true, 16384, "append mix, in place, U=2, 16384 blocks", 16.0);
    RUN(1, 15, true, 262144, "append mix, in place, U=1, one tile per block", 16.0);
    RUN(2, 15, false, 4096, "append mix, top to other buffer, U=2", 16.0);
    RUN(2, 2, true, 4096, "R len+top, W top only (9 B/col)", 9.0);
    RUN(2, 3, true, 4096, "R len+top, W plane+top (13 B/col)", 13.0);
    RUN(2, 7, true, 4096, "R len+top, W plane+top+z0 (15 B/col)", 15.0);
    RUN(2, 1, true, 4096, "R len+top, W plane (9 B/col)", 9.0);
    RUN(2, 0, true, 4096, "R len+top only (5 B/col)", 5.0);
    uint32_t* planes; CHK(hipMalloc(&planes, 4 * n * 4)); CHK(hipMemset(planes, 1, 4 * n * 4));
#define RUNS(NPL, RUNLEN, HALF, NAME) time([&] { hipLaunchKernelGGL((k_scatter<NPL>), dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, len, top, z0, planes, n, RUNLEN, HALF); }, NAME, 16.0)
    RUNS(1, 1, 0, "append, one plane");
    RUNS(2, 64, 0, "append, plane alternates every 256 columns");
    RUNS(2, 8, 0, "append, plane alternates every 32 columns (128 B)");
    RUNS(2, 4, 0, "append, plane alternates every 16 columns (64 B)");
    RUNS(2, 2, 0, "append, plane alternates every 8 columns (32 B)");
    RUNS(2, 1, 0, "append, plane alternates every 4 columns (16 B)");
    RUNS(4, 2, 0, "append, 4 planes, every 8 columns");
    RUNS(2, 1, 1, "append, plane alternates every 2 columns (8 B)");
    return 0;
}
