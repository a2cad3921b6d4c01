// Test infrastructure (host only, runs without a GPU): drives the library's launch guard -- kc_grid / KC_LAUNCH / KC_HIP of
// kmercamel_amd/csrc/kc_internal.h -- through a 1-D launch of 2^32 threads and checks that the CALLING function returns KC_E_ARG before
// anything is launched, and that the overflow flag never outlives the check that consumed it.
#include "../../kmercamel_amd/csrc/kc_internal.h"

__global__ void grid_check_dummy_kernel(int *) {}

static int launch_n(kc_ctx *ctx, u64 n, bool *reached_after) {
    KC_LAUNCH(grid_check_dummy_kernel, dim3(kc_grid(n, 256)), dim3(256), 0, ctx->stream, (int *)nullptr);
    *reached_after = true;   // (only with a GPU: not exercised here for small n)
    return KC_OK;
}
static int hip_status(kc_ctx *ctx, hipError_t e) { KC_HIP(ctx, e); return KC_OK; }
static int grid_then_status(kc_ctx *ctx, u64 n, hipError_t e) { unsigned g = kc_grid(n, 256); (void)g; KC_HIP(ctx, e); return KC_OK; }

#define CHECK(cond) do { if (!(cond)) { printf("FAIL line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

int main() {
    kc_ctx c;
    bool after = false;
    // a launch of exactly 2^32 threads is refused at the call site, nothing runs behind it
    CHECK(launch_n(&c, (u64)1 << 32, &after) == KC_E_ARG);
    CHECK(!after);
    CHECK(c.err.find("2^32") != std::string::npos);
    CHECK(!kc_grid_overflowed);
    // ... so a later, unrelated HIP error is reported as what it is
    CHECK(hip_status(&c, hipErrorOutOfMemory) == KC_E_NO_DEVICE);
    CHECK(c.err.find("2^32") == std::string::npos);
    // an overflowing kc_grid whose value reaches no launch is caught by the next KC_HIP, even a successful one, and only by that one
    CHECK(grid_then_status(&c, ((u64)1 << 32) + 5, hipSuccess) == KC_E_ARG);
    CHECK(hip_status(&c, hipSuccess) == KC_OK);
    // the largest launch that fits and ordinary sizes do not trip it
    CHECK(kc_grid(((u64)1 << 32) - 256, 256) == (1u << 24) - 1 && !kc_grid_overflowed);
    CHECK(kc_grid(1000, 256) == 4 && kc_grid(0, 256) == 1 && !kc_grid_overflowed);
    CHECK(hip_status(&c, hipSuccess) == KC_OK);
    printf("grid_check ok\n");
    return 0;
}
