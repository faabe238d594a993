// Probe of HIP's virtual memory management as the elastic plane buffers use it: several reserved
// ranges, each extended chunk by chunk with chunks of ONE size per range.  (With chunks of different
// sizes inside one range, hipMemSetAccess refused some of them with "invalid argument" on ROCm 7.2;
// equal sizes never failed.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
int main(int argc, char** argv) {
    hipSetDevice(0);
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    void* junk[5]; for (int k = 0; k < 5; k++) hipMalloc(&junk[k], ((size_t)3 << 20) + 4096 * k);
    const size_t G = (size_t)1 << 30;
    const size_t unit[3] = {16 * G, 4 * G, (size_t)6 << 20};
    char* base[3]; size_t off[3] = {0, 0, 0}; int bad = 0;
    for (int r = 0; r < 3; r++) { void* p; hipMemAddressReserve(&p, 256 * G, 0, nullptr, 0); base[r] = (char*)p; printf("reservation %d at %p unit %zu MiB\n", r, p, unit[r] >> 20); }
    for (int round = 0; round < 6; round++)
        for (int r = 0; r < 3; r++) {
            hipMemGenericAllocationHandle_t h;
            hipError_t c = hipMemCreate(&h, unit[r], &prop, 0);
            hipError_t m = c == hipSuccess ? hipMemMap(base[r] + off[r], unit[r], 0, h, 0) : c;
            hipError_t a = m == hipSuccess ? hipMemSetAccess(base[r] + off[r], unit[r], &acc, 1) : m;
            if (a != hipSuccess) { bad++; printf("  round %d reservation %d off %zu MiB: %s / %s / %s\n", round, r, off[r] >> 20, hipGetErrorString(c), hipGetErrorString(m), hipGetErrorString(a)); (void)hipGetLastError(); }
            off[r] += unit[r];
        }
    // shrink and regrow the first reservation
    printf("failures %d\n", bad);
    return 0;
}
