// What does a HIP process cost before main and after _exit?  (round 5; run on the GPU box through tools/exit_probe.py)  Not product code.
//   hipcc -O2 --offload-arch=gfx950 -o tools/exit_probe.bin tools/exit_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>
static double epoch() { struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
__global__ void k(int *p) { if (p) *p = 1; }
int main(int argc, char **argv) {
    const double t_main = epoch();
    const int mode = argc > 1 ? atoi(argv[1]) : 0;      // 0: nothing; 1: hipInit + stream; 2: + kernel launch; 3: + 120 GB hipMalloc (untouched); 4: + touched (memset)
    hipStream_t s = nullptr;
    char *d = nullptr;
    if (mode >= 1) { (void)hipInit(0); (void)hipSetDevice(0); (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
    if (mode >= 2) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, (int *)nullptr); (void)hipStreamSynchronize(s); }
    if (mode >= 3) { (void)hipMalloc((void **)&d, (size_t)120 << 30); }
    if (mode >= 4) { (void)hipMemsetAsync(d, 1, (size_t)120 << 30, s); (void)hipStreamSynchronize(s); }
    const double t_exit = epoch();
    printf("%.6f %.6f\n", t_main, t_exit);
    fflush(stdout);
    if (argc > 2 && atoi(argv[2])) return 0;   // ordinary exit (runtime tear-down) instead of _exit
    _exit(0);
}
