// probe 2: can the 37 ms/GB of large hipMalloc calls be overlapped (threads), avoided (stream-ordered pool, VMM API) or amortised?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipFree(0);
    const size_t G20 = (size_t)20e9;
    for (int nt : {1, 2, 4, 8}) {
        std::vector<void *> p(nt, nullptr);
        std::vector<std::thread> th;
        double t0 = now();
        for (int i = 0; i < nt; ++i) th.emplace_back([&, i] { hipSetDevice(0); hipMalloc(&p[i], G20); });
        for (auto &t : th) t.join();
        double t1 = now();
        for (auto q : p) hipFree(q);
        printf("%d threads x 20 GB hipMalloc in parallel: %.3f s (%.1f ms/GB), free %.3f s\n", nt, t1 - t0, (t1 - t0) * 1e3 / (20.0 * nt), now() - t1);
    }
    {
        void *p = nullptr; double t0 = now(); hipMalloc(&p, G20); double t1 = now(); hipFree(p); double t2 = now();
        void *q = nullptr; hipMalloc(&q, G20); double t3 = now(); hipFree(q);
        printf("20 GB malloc %.3f, free %.3f, malloc again %.3f\n", t1 - t0, t2 - t1, t3 - t2);
    }
    {
        hipStream_t s; hipStreamCreate(&s);
        void *p = nullptr; double t0 = now(); hipError_t e = hipMallocAsync(&p, G20, s); hipStreamSynchronize(s); double t1 = now();
        if (e == hipSuccess) { hipMemsetAsync(p, 0, G20, s); hipStreamSynchronize(s); }
        double t2 = now();
        hipFreeAsync(p, s); hipStreamSynchronize(s); double t3 = now();
        void *q = nullptr; hipMallocAsync(&q, G20, s); hipStreamSynchronize(s); double t4 = now();
        printf("hipMallocAsync 20 GB: %.3f s (%d), memset %.3f, freeAsync %.3f, mallocAsync again %.3f\n", t1 - t0, (int)e, t2 - t1, t3 - t2, t4 - t3);
        hipFreeAsync(q, s); hipStreamSynchronize(s);
    }
    {
        size_t gran = 0;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        size_t n = (G20 + gran - 1) / gran * gran;
        hipMemGenericAllocationHandle_t h;
        double t0 = now(); hipError_t e1 = hipMemCreate(&h, n, &prop, 0); double t1 = now();
        void *va = nullptr; hipError_t e2 = hipMemAddressReserve(&va, n, 0, nullptr, 0); double t2 = now();
        hipError_t e3 = hipMemMap(va, n, 0, h, 0); double t3 = now();
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        hipError_t e4 = hipMemSetAccess(va, n, &acc, 1); double t4 = now();
        if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess && e4 == hipSuccess) { hipMemset(va, 0, n); hipDeviceSynchronize(); }
        double t5 = now();
        printf("VMM 20 GB (granularity %zu): create %.3f (%d) reserve %.3f (%d) map %.3f (%d) setaccess %.3f (%d) memset %.3f\n", gran, t1 - t0, (int)e1, t2 - t1, (int)e2, t3 - t2, (int)e3, t4 - t3, (int)e4, t5 - t4);
    }
    {
        void *p = nullptr; double t0 = now(); hipError_t e = hipExtMallocWithFlags(&p, G20, hipDeviceMallocUncached); double t1 = now();
        printf("hipExtMallocWithFlags(uncached) 20 GB %.3f (%d)\n", t1 - t0, (int)e); hipFree(p);
    }
    {
        void *p = nullptr; double t0 = now(); hipError_t e = hipMallocManaged(&p, G20); double t1 = now();
        double t2 = now();
        printf("hipMallocManaged 20 GB %.3f (%d), prefetch to device %.3f\n", t1 - t0, (int)e, t2 - t1); hipFree(p);
    }
    return 0;
}
