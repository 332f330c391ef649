// rows4_max / rows4_sum (vq_common.h: v_permlane16_swap + v_permlane32_swap) against the __shfl_xor butterfly they replace.
// build: hipcc --offload-arch=gfx950 -O3 -I video-quierer_amd/csrc scripts/ubench/rows4_check.hip -o /tmp/rows4_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "vq_common.h"
__global__ void k(const float* in, float* o) {
    const int l = threadIdx.x;
    float x = in[l];
    float m = x; m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    float s = x; s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    o[l] = m; o[64 + l] = s; o[128 + l] = vq::rows4_max(x); o[192 + l] = vq::rows4_sum(x);
}
int main() {
    float h[64], r[256]; for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 61) + 0.25f * (i >> 4);
    float *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(r));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) if (r[i] != r[128 + i] || r[64 + i] != r[192 + i]) { if (bad < 8) printf("lane %d: shfl max %g sum %g, rows4 max %g sum %g\n", i, r[i], r[64 + i], r[128 + i], r[192 + i]); ++bad; }
    printf("%d lanes differ\n", bad);
    return bad != 0;
}
