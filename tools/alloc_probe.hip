// probe: what do large hipMalloc / hipFree / pageable vs pinned copies cost on this box?  (for the CLI's wall time)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; if (i < n) p[i] = 1; }
int main() {
    double t0 = now();
    hipFree(0);
    printf("runtime init %.3f s\n", now() - t0);
    for (double gb : {1.0, 8.0, 20.0, 41.0, 82.0}) {
        size_t n = (size_t)(gb * 1e9);
        void *p = nullptr;
        t0 = now();
        hipError_t e = hipMalloc(&p, n);
        double t1 = now();
        hipMemset(p, 0, n); hipDeviceSynchronize();
        double t2 = now();
        hipMemset(p, 1, n); hipDeviceSynchronize();
        double t3 = now();
        hipFree(p);
        double t4 = now();
        printf("%.0f GB: malloc %.3f s (%d), first memset %.3f s, second memset %.3f s, free %.3f s\n", gb, t1 - t0, (int)e, t2 - t1, t3 - t2, t4 - t3);
    }
    size_t n = (size_t)2600 << 20;
    char *h = (char *)malloc(n); memset(h, 65, n);
    char *hp = nullptr; 
    t0 = now(); hipHostMalloc((void **)&hp, n, hipHostMallocDefault); printf("hipHostMalloc 2.6 GB %.3f s\n", now() - t0);
    t0 = now(); memcpy(hp, h, n); printf("host memcpy 2.6 GB %.3f s\n", now() - t0);
    char *d = nullptr; hipMalloc((void **)&d, n);
    t0 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); printf("H2D pageable 2.6 GB %.3f s\n", now() - t0);
    t0 = now(); hipMemcpy(d, hp, n, hipMemcpyHostToDevice); printf("H2D pinned 2.6 GB %.3f s\n", now() - t0);
    t0 = now(); hipMemcpy(h, d, n, hipMemcpyDeviceToHost); printf("D2H pageable 2.6 GB %.3f s\n", now() - t0);
    t0 = now(); hipMemcpy(hp, d, n, hipMemcpyDeviceToHost); printf("D2H pinned 2.6 GB %.3f s\n", now() - t0);
    t0 = now(); hipHostRegister(h, n, hipHostRegisterDefault); printf("hipHostRegister 2.6 GB %.3f s\n", now() - t0);
    t0 = now(); hipMemcpy(d, h, n, hipMemcpyHostToDevice); printf("H2D registered 2.6 GB %.3f s\n", now() - t0);
    FILE *f = fopen("/dev/shm/kc_probe.bin", "wb"); t0 = now(); fwrite(h, 1, n, f); fclose(f); printf("write 2.6 GB to /dev/shm %.3f s\n", now() - t0);
    f = fopen("/dev/shm/kc_probe.bin", "rb"); t0 = now(); size_t got = fread(hp, 1, n, f); fclose(f); printf("read 2.6 GB from /dev/shm into pinned %.3f s (%zu)\n", now() - t0, got);
    remove("/dev/shm/kc_probe.bin");
    return 0;
}
