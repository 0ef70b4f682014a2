// Layout probe for v_mfma_f32_4x4x1_16b_f32 on gfx950: prints which (a-lane, b-lane) product lands in which (lane, register).
// hipcc --offload-arch=gfx950 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe.bin && ./tools/mfma4x4_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* o) {
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) o[4 * threadIdx.x + i] = c[i];
}
int main() {
    float ha[64], hb[64], ho[256];
    // a = distinct primes-ish per lane, b = powers: product identifies the pair
    for (int l = 0; l < 64; ++l) { ha[l] = (float)(l + 1); hb[l] = 1.f + 100.f * l; }
    float *a, *b, *o;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const long v = (long)(ho[4 * l + i] + 0.5f);           // (1 + x)(1 + 100 y), x < 64: x = v mod 100 - 1
            const int fa = (int)(v % 100) - 1, fb = (int)((v / (fa + 1) - 1) / 100);
            if (i == 0) printf("lane %2d:", l);
            printf("  r%d=a[%2d]*b[%2d]", i, fa, fb);
            if (i == 3) printf("\n");
            if (fa != 4 * (l >> 2) + i || fb != l) ok = 0;
        }
    printf("expected layout (lane 4b+j, reg i) = a[4b+i] * b[4b+j]: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    return 0;
}
