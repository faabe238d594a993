// Micro-benchmarks of the access patterns the column kernels use, to see what each reaches on HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

template <typename T, int U>
__global__ __launch_bounds__(256) void k_copy(const T* __restrict__ in, T* __restrict__ out, size_t n) {
    const size_t tile = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * tile + threadIdx.x; base < n; base += (size_t)gridDim.x * tile) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t i = base + (size_t)u * 256; if (i < n) v[u] = in[i]; }
#pragma unroll
        for (int u = 0; u < U; u++) { size_t i = base + (size_t)u * 256; if (i < n) out[i] = v[u]; }
    }
}

// the deposit-extend pattern: len[c] (u8) -> z1[(len-1)*P + c] (u16) read-modify-write, mat read
template <int U>
__global__ __launch_bounds__(256) void k_extend(const uint8_t* __restrict__ len, const uint8_t* __restrict__ mat, uint16_t* __restrict__ z1, size_t n, size_t P, int matId) {
    const size_t tile = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * tile + threadIdx.x; base < n; base += (size_t)gridDim.x * tile) {
        int l[U]; uint16_t q[U]; int m[U];
#pragma unroll
        for (int u = 0; u < U; u++) { size_t c = base + (size_t)u * 256; l[u] = c < n ? len[c] : 0; }
#pragma unroll
        for (int u = 0; u < U; u++) { size_t c = base + (size_t)u * 256; q[u] = 0; m[u] = -1; if (l[u] > 0) { size_t t = (size_t)(l[u] - 1) * P + c; q[u] = z1[t]; m[u] = mat[t]; } }
#pragma unroll
        for (int u = 0; u < U; u++) { size_t c = base + (size_t)u * 256; if (l[u] > 0 && m[u] == matId) z1[(size_t)(l[u] - 1) * P + c] = q[u] + 100; }
    }
}

int main(int argc, char** argv) {
    const size_t n = (size_t)16384 * 16384;
    const int L = 8;
    uint8_t *len, *mat; uint16_t *z1, *z1b;
    for (int pad = 0; pad < 2; pad++) {
        const size_t P = n + (pad ? 256 * 33 : 0);
        CHK(hipMalloc(&len, n)); CHK(hipMalloc(&mat, P * L)); CHK(hipMalloc(&z1, P * L * 2)); CHK(hipMalloc(&z1b, P * L * 2));
        CHK(hipMemset(len, 3, n)); CHK(hipMemset(mat, 4, P * L)); CHK(hipMemset(z1, 1, P * L * 2));
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        auto time = [&](auto launch, const char* name, double bytes) {
            launch(); hipDeviceSynchronize();
            float best = 1e9;
            for (int r = 0; r < 5; r++) { hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
            printf("pad=%d %-44s %8.3f ms  %8.1f GB/s\n", pad, name, best, bytes / best / 1e6);
        };
        const int G = 4096;
        time([&] { hipLaunchKernelGGL((k_copy<uint16_t, 1>), dim3(G), dim3(256), 0, 0, z1, z1b, n); }, "copy u16 1/thread", 4.0 * n);
        time([&] { hipLaunchKernelGGL((k_copy<uint16_t, 8>), dim3(G), dim3(256), 0, 0, z1, z1b, n); }, "copy u16 8/thread (strided 256)", 4.0 * n);
        time([&] { hipLaunchKernelGGL((k_copy<u16x8, 1>), dim3(G), dim3(256), 0, 0, (const u16x8*)z1, (u16x8*)z1b, n / 8); }, "copy u16x8 (16 B/lane)", 4.0 * n);
        time([&] { hipLaunchKernelGGL((k_copy<u16x8, 4>), dim3(G), dim3(256), 0, 0, (const u16x8*)z1, (u16x8*)z1b, n / 8); }, "copy u16x8 4/thread", 4.0 * n);
        time([&] { hipLaunchKernelGGL((k_copy<uint8_t, 8>), dim3(G), dim3(256), 0, 0, mat, mat + P, n); }, "copy u8 8/thread", 2.0 * n);
        time([&] { hipLaunchKernelGGL((k_copy<uint32_t, 8>), dim3(G), dim3(256), 0, 0, (const uint32_t*)z1, (uint32_t*)z1b, n / 2); }, "copy u32 8/thread", 4.0 * n);
        time([&] { hipLaunchKernelGGL((k_extend<1>), dim3(G), dim3(256), 0, 0, len, mat, z1, n, P, 4); }, "extend pattern U=1 (6 B/col)", 6.0 * n);
        time([&] { hipLaunchKernelGGL((k_extend<4>), dim3(G), dim3(256), 0, 0, len, mat, z1, n, P, 4); }, "extend pattern U=4", 6.0 * n);
        time([&] { hipLaunchKernelGGL((k_extend<8>), dim3(G), dim3(256), 0, 0, len, mat, z1, n, P, 4); }, "extend pattern U=8", 6.0 * n);
        time([&] { hipLaunchKernelGGL((k_extend<8>), dim3(16384), dim3(256), 0, 0, len, mat, z1, n, P, 4); }, "extend pattern U=8 grid 16384", 6.0 * n);
        time([&] { hipLaunchKernelGGL((k_extend<16>), dim3(G), dim3(256), 0, 0, len, mat, z1, n, P, 4); }, "extend pattern U=16", 6.0 * n);
        hipFree(len); hipFree(mat); hipFree(z1); hipFree(z1b);
    }
    return 0;
}
