// Probe of the HIP virtual-memory API on this driver: which map / set-access sequences are accepted.
// API calls only -- mapped memory is never touched, so a refused call cannot turn into a GPU fault.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static hipError_t ck(const char *what, hipError_t e)
{
    printf("    %-34s -> %s\n", what, hipGetErrorName(e));
    fflush(stdout);
    (void)hipGetLastError();
    return e;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    ck("granularity(min)", hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    ck("granularity(rec)", hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity min %zu recommended %zu\n", gmin, grec);
    const size_t MB = 1 << 20, GB = (size_t)1 << 30;
    struct Case { size_t a, b, align_b; int whole; };
    const Case cases[] = {
        {2 * MB, 2 * MB, 0, 0}, {2 * MB, 2 * MB, 0, 1}, {294 * MB, 294 * MB, 0, 0}, {294 * MB, 294 * MB, 0, 1},
        {3 * GB, 3 * GB, 0, 0}, {1 * GB, 1 * GB, 0, 0}, {512 * MB, 512 * MB, 0, 0}, {294 * MB, 294 * MB, GB, 0},
        {294 * MB, 294 * MB, 512 * MB, 0}, {256 * MB, 256 * MB, 0, 0}, {64 * MB, 64 * MB, 0, 0}, {2 * MB, 64 * MB, 64 * MB, 0},
    };
    for (const Case &c : cases) {
        printf("case first %zu MB, second %zu MB, second offset aligned to %zu MB, set-access on %s\n", c.a / MB, c.b / MB,
               c.align_b / MB, c.whole ? "[base, end)" : "the new chunk");
        void *base = nullptr;
        const size_t reserve = 16 * GB;
        if (ck("reserve", hipMemAddressReserve(&base, reserve, c.align_b ? c.align_b : grec, nullptr, 0)) != hipSuccess) continue;
        hipMemGenericAllocationHandle_t h1, h2;
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        bool m1 = false, m2 = false, c1 = false, c2 = false;
        size_t off2 = c.a;
        if (c.align_b) off2 = (off2 + c.align_b - 1) / c.align_b * c.align_b;
        do {
            if (ck("create 1", hipMemCreate(&h1, c.a, &prop, 0)) != hipSuccess) break;
            c1 = true;
            if (ck("map 1", hipMemMap(base, c.a, 0, h1, 0)) != hipSuccess) break;
            m1 = true;
            if (ck("set access 1", hipMemSetAccess(base, c.a, &acc, 1)) != hipSuccess) break;
            if (ck("create 2", hipMemCreate(&h2, c.b, &prop, 0)) != hipSuccess) break;
            c2 = true;
            if (ck("map 2", hipMemMap((char *)base + off2, c.b, 0, h2, 0)) != hipSuccess) break;
            m2 = true;
            if (c.whole) ck("set access [base,end)", hipMemSetAccess(base, off2 + c.b, &acc, 1));
            else ck("set access 2", hipMemSetAccess((char *)base + off2, c.b, &acc, 1));
        } while (0);
        if (m2) ck("unmap 2", hipMemUnmap((char *)base + off2, c.b));
        if (c2) ck("release 2", hipMemRelease(h2));
        if (m1) ck("unmap 1", hipMemUnmap(base, c.a));
        if (c1) ck("release 1", hipMemRelease(h1));
        ck("address free", hipMemAddressFree(base, reserve));
    }
    return 0;
}
