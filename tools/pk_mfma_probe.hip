// Minimal two-kernel check of the round-2 finding behind csrc/Makefile's NOPACK rule: does a wave that executes packed
// fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) compute wrong values while ANOTHER kernel's waves issue
// v_mfma_f32_32x32x16_bf16 on the same CUs?  No library code: a victim kernel whose every lane runs a chain of packed operations
// with an exactly known result, and aggressor kernels that only issue MFMAs (bf16 32x32x16, and fp32 32x32x2 as the control).
// Each configuration: victim alone, victim beside the aggressor on a second stream, REPS times; prints the number of lanes with a
// wrong result.  A non-zero count beside the bf16 aggressor and zero otherwise reproduces the finding; zeros everywhere mean this
// probe does not (the attribution in DESIGN 3.4 then stays unproven).
//   hipcc --offload-arch=gfx950 -O2 tools/pk_mfma_probe.hip -o tools/pk_mfma_probe.bin && ./tools/pk_mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// victim: x <- x * 1 + 1 (packed), y <- y * 1 (packed mul), z <- z + 1 (packed add), ITERS times, on integer-valued floats: exact.
__global__ void victim(float* out, int iters)
{
    const float base = (float)(threadIdx.x & 63);
    f2 x = {base, base + 1.f}, y = {base + 2.f, base + 3.f}, z = {base + 4.f, base + 5.f};
    const f2 one = {1.f, 1.f};
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(one));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y) : "v"(one));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(z) : "v"(one));
    }
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[6 * t + 0] = x.x; out[6 * t + 1] = x.y; out[6 * t + 2] = y.x; out[6 * t + 3] = y.y; out[6 * t + 4] = z.x; out[6 * t + 5] = z.y;
}

__global__ void aggressor_bf16(float* sink, int iters)
{
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.001f * threadIdx.x); b[i] = (__bf16)(0.5f); }
    f16v acc = {0};
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1];
}
__global__ void aggressor_f32(float* sink, int iters)
{
    f16v acc = {0};
    const float a = 1.0f + 0.001f * threadIdx.x, b = 0.5f;
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1];
}

int main()
{
    const int WGS = 4096, TPB = 256, VIT = 200000, REPS = 8;
    float *out, *sink;
    CK(hipMalloc(&out, sizeof(float) * 6 * (size_t)WGS * TPB));
    CK(hipMalloc(&sink, sizeof(float) * 1024));
    std::vector<float> h(6 * (size_t)WGS * TPB);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto count_wrong = [&]() {
        long wrong = 0;
        for (size_t t = 0; t < (size_t)WGS * TPB; ++t) {
            const float base = (float)(t & 63);
            const float want[6] = {base + VIT, base + 1.f + VIT, base + 2.f, base + 3.f, base + 4.f + VIT, base + 5.f + VIT};
            for (int k = 0; k < 6; ++k) wrong += h[6 * t + k] != want[k];
        }
        return wrong;
    };
    const char* names[3] = {"victim alone", "victim beside v_mfma_f32_32x32x16_bf16", "victim beside v_mfma_f32_32x32x2_f32"};
    for (int cfg = 0; cfg < 3; ++cfg) {
        long total = 0; float ms_sum = 0.f;
        for (int rep = 0; rep < REPS; ++rep) {
            CK(hipMemsetAsync(out, 0, sizeof(float) * 6 * (size_t)WGS * TPB, s1));
            CK(hipDeviceSynchronize());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            if (cfg == 1) hipLaunchKernelGGL(aggressor_bf16, dim3(2048), dim3(256), 0, s2, sink, 60000);
            if (cfg == 2) hipLaunchKernelGGL(aggressor_f32, dim3(2048), dim3(256), 0, s2, sink, 60000);
            CK(hipEventRecord(a, s1));
            hipLaunchKernelGGL(victim, dim3(WGS), dim3(TPB), 0, s1, out, VIT);
            CK(hipEventRecord(b, s1));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms_sum += ms;
            CK(hipMemcpy(h.data(), out, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
            total += count_wrong();
            CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
        }
        printf("%-44s: %ld wrong values in %d repetitions (victim %.1f ms per launch)\n", names[cfg], total, REPS, ms_sum / REPS);
    }
    return 0;
}
