// Micro-benchmark 4: stream layouts for an "append" in the chained form (dense plane per step), as
// the layout stood when the top's bottom (topz0) and len were still streams of their own.
//  A: separate small arrays  R len1 link1 top4 | W rec4 prv1 topz0_2 len1 link1 top4          (19 B/col, 9 streams)
//  B: len+link packed u16    R ll2 top4        | W rec4 prv1 topz0_2 ll2 top4                  (19 B/col, 7 streams)
//  C: aux u32 (z0|link|len)  R aux4 top4       | W rec4 prv1 aux4 top4                         (21 B/col, 6 streams)
//  D: as C, prv inside a u64 record plane       R aux4 top4 | W rec8 aux4 top4                   (24 B/col, 5 streams)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t* len, uint8_t* link, uint16_t* ll, uint32_t* aux, uint32_t* top, uint16_t* z0, uint32_t* rec, uint8_t* prv, uint32_t* rec8, size_t n, int P) {
    const size_t ng = n / 4;
    const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= ng) return;
    const size_t c0 = gi * 4;
    u32x4 tv = *(const u32x4*)(top + c0);
    u32x4 nv = tv + 100u;
    if (MODE == 0) {
        u8x4 ln = *(const u8x4*)(len + c0), lk = *(const u8x4*)(link + c0);
        *(u32x4*)(rec + c0) = tv; *(u8x4*)(prv + c0) = lk;
        u16x4 z; z[0] = tv[0]; z[1] = tv[1]; z[2] = tv[2]; z[3] = tv[3];
        *(u16x4*)(z0 + c0) = z; *(u8x4*)(len + c0) = ln + (unsigned char)1; u8x4 pk = {(unsigned char)P, (unsigned char)P, (unsigned char)P, (unsigned char)P};
        *(u8x4*)(link + c0) = pk; *(u32x4*)(top + c0) = nv;
    } else if (MODE == 1) {
        u16x4 l2 = *(const u16x4*)(ll + c0);
        u8x4 lk; lk[0] = l2[0] >> 8; lk[1] = l2[1] >> 8; lk[2] = l2[2] >> 8; lk[3] = l2[3] >> 8;
        *(u32x4*)(rec + c0) = tv; *(u8x4*)(prv + c0) = lk;
        u16x4 z; z[0] = tv[0]; z[1] = tv[1]; z[2] = tv[2]; z[3] = tv[3];
        *(u16x4*)(z0 + c0) = z;
        u16x4 n2; for (int j = 0; j < 4; j++) n2[j] = (unsigned short)(((l2[j] & 255) + 1) | (P << 8));
        *(u16x4*)(ll + c0) = n2; *(u32x4*)(top + c0) = nv;
    } else if (MODE == 2) {
        u32x4 a = *(const u32x4*)(aux + c0);
        u8x4 lk; for (int j = 0; j < 4; j++) lk[j] = (a[j] >> 16) & 255;
        *(u32x4*)(rec + c0) = tv; *(u8x4*)(prv + c0) = lk;
        u32x4 na; for (int j = 0; j < 4; j++) na[j] = (tv[j] & 0xffff) | ((unsigned)P << 16) | (((a[j] >> 24) + 1) << 24);
        *(u32x4*)(aux + c0) = na; *(u32x4*)(top + c0) = nv;
    } else {
        u32x4 a = *(const u32x4*)(aux + c0);
        u32x4 lo, hi; lo[0] = tv[0]; lo[1] = a[0] >> 16; lo[2] = tv[1]; lo[3] = a[1] >> 16; hi[0] = tv[2]; hi[1] = a[2] >> 16; hi[2] = tv[3]; hi[3] = a[3] >> 16;
        *(u32x4*)(rec8 + 2 * c0) = lo; *(u32x4*)(rec8 + 2 * c0 + 4) = hi;
        u32x4 na; for (int j = 0; j < 4; j++) na[j] = (tv[j] & 0xffff) | ((unsigned)P << 16) | (((a[j] >> 24) + 1) << 24);
        *(u32x4*)(aux + c0) = na; *(u32x4*)(top + c0) = nv;
    }
}

int main() {
    const size_t n = (size_t)16384 * 16384;
    uint8_t *len, *link, *prv; uint16_t *ll, *z0; uint32_t *aux, *top, *rec, *rec8;
    CHK(hipMalloc(&len, n)); CHK(hipMalloc(&link, n)); CHK(hipMalloc(&prv, n)); CHK(hipMalloc(&ll, 2 * n)); CHK(hipMalloc(&z0, 2 * n));
    CHK(hipMalloc(&aux, 4 * n)); CHK(hipMalloc(&top, 4 * n)); CHK(hipMalloc(&rec, 4 * n)); CHK(hipMalloc(&rec8, 8 * n));
    CHK(hipMemset(len, 1, n)); CHK(hipMemset(link, 1, n)); CHK(hipMemset(ll, 1, 2 * n)); CHK(hipMemset(aux, 1, 4 * n)); CHK(hipMemset(top, 1, 4 * n));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    auto time = [&](auto launch, const char* name, double bytes) {
        launch(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
        printf("%-60s %8.3f ms  %8.1f GB/s\n", name, best, bytes * n / best / 1e6);
    };
    const unsigned G = (unsigned)(n / 4 / 256);
#define RUN(M, NAME, BYTES) time([&] { hipLaunchKernelGGL((k<M>), dim3(G), dim3(256), 0, 0, len, link, ll, aux, top, z0, rec, prv, rec8, n, 3); }, NAME, BYTES)
    RUN(0, "A separate len/link/topz0/prv (19 B/col, 9 streams)", 19.0);
    RUN(1, "B len+link packed u16 (19 B/col, 7 streams)", 19.0);
    RUN(2, "C aux u32 = z0|link|len (21 B/col, 6 streams)", 21.0);
    RUN(3, "D aux u32, u64 records (24 B/col, 5 streams)", 24.0);
    return 0;
}
