// OCP e4m3 conversions on gfx950 (v_cvt_pk_fp8_f32 / v_cvt_pk_f32_fp8) as the split residual stream uses them: values beyond +-448 convert to NaN
// (hence the clamp in pack4_fp8), 2^-9 is the smallest subnormal.   hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/fp8_probe.bin scripts/ubench/fp8_probe.hip
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int* out, float* back) {
    const int i = threadIdx.x;
    float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    out[i] = w;
    f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    back[4 * i] = lo[0]; back[4 * i + 1] = lo[1]; back[4 * i + 2] = hi[0]; back[4 * i + 3] = hi[1];
}
int main() {
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (i - 128) * 0.37f * (i % 7 == 0 ? 40.f : 1.f);
    h[3] = 500.f; h[5] = -1000.f; h[9] = 1e-4f; h[11] = 448.f; h[13] = 0.0019f;
    float *di, *db; int* dout; hipMalloc(&di, sizeof(h)); hipMalloc(&db, sizeof(h)); hipMalloc(&dout, 64 * 4);
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, db);
    float r[256]; hipMemcpy(r, db, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < 24; ++i) printf("%g -> %g\n", h[i], r[i]);
    return 0;
}
