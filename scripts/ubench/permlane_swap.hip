// What v_permlane16_swap / v_permlane32_swap do on gfx950: both operands carry the lane id (+ 100 for the second).
// build: hipcc --offload-arch=gfx950 -O2 scripts/ubench/permlane_swap.hip -o /tmp/permlane_swap && /tmp/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    const unsigned l = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    o[l] = a[0]; o[64 + l] = a[1]; o[128 + l] = b[0]; o[192 + l] = b[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap result[0]", "permlane16_swap result[1]", "permlane32_swap result[0]", "permlane32_swap result[1]"};
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
