// Ablation probe for the fp32 MFMA NT GEMM main loop (not part of the library).
// build: hipcc -O3 --offload-arch=gfx950 tools/gemm_probe.hip -o gpurun_out/gemm_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// MODE 0 full, 1 no global loads in loop, 2 no global loads + no LDS stores, 3 = 2 + no barrier, 4 MFMA only
template <int MODE, int BK, int WPS>
__global__ __launch_bounds__(256, WPS)
void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int tiles_m, int tiles_n)
{
    constexpr int BM = 128, BN = 128, LDT = BK + 4, KQ = BK / 4, RPP = 256 / KQ, NP = BM / RPP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];
    float* As = smem; float* Bs = smem + 2 * BM * LDT;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int tn = slot % tiles_n, tm = (slot / tiles_n) * 8 + xcd;
    if (tm >= tiles_m) return;
    const int t = threadIdx.x, lr = t / KQ, kq = (t % KQ) * 4;
    const int row_base = tm * BM, col_base = tn * BN;
    const float* arow[NP]; const float* brow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        arow[p] = A + (size_t)min(row_base + lr + RPP * p, M - 1) * K;
        brow[p] = B + (size_t)min(col_base + lr + RPP * p, N - 1) * K;
    }
    const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 ra4[NP], rb4[NP];
    auto g_load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { ra4[p] = ldg4(arow[p] + k0 + kq); rb4[p] = ldg4(brow[p] + k0 + kq); }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            stg4(As + buf * BM * LDT + (lr + RPP * p) * LDT + kq, ra4[p]);
            stg4(Bs + buf * BN * LDT + (lr + RPP * p) * LDT + kq, rb4[p]);
        }
    };
    const int nk = K / BK;
    g_load(0); s_store(0); s_store(1);
    __syncthreads();
    float4 fa0 = ldg4(As + (wm * 64 + l31) * LDT + 4 * h), fb0 = ldg4(Bs + (wn * 64 + l31) * LDT + 4 * h);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (MODE == 0 && kt + 1 < nk) g_load((kt + 1) * BK);
        const float* ab = As + cur * BM * LDT + (wm * 64 + l31) * LDT + 4 * h;
        const float* bb = Bs + cur * BN * LDT + (wn * 64 + l31) * LDT + 4 * h;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            float4 a0, a1, b0, b1;
            if (MODE == 4) { a0 = fa0; a1 = fa0; b0 = fb0; b1 = fb0; }
            else { a0 = ldg4(ab + kg * 8); a1 = ldg4(ab + 32 * LDT + kg * 8); b0 = ldg4(bb + kg * 8); b1 = ldg4(bb + 32 * LDT + kg * 8); }
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][0] = mfma32(av0[q], bv0[q], acc[0][0]); acc[0][1] = mfma32(av0[q], bv1[q], acc[0][1]);
                acc[1][0] = mfma32(av1[q], bv0[q], acc[1][0]); acc[1][1] = mfma32(av1[q], bv1[q], acc[1][1]);
            }
        }
        if (MODE <= 1 && kt + 1 < nk) s_store(cur ^ 1);
        if (MODE <= 2) __syncthreads();
    }
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int r = 0; r < 16; ++r) {
        const int row = row_base + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = col_base + wn * 64 + ni * 32 + l31;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[mi][ni][r];
    }
}

template <int MODE, int BK, int WPS>
float run(const float* A, const float* B, float* C, int M, int N, int K)
{
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    const int blocks = (tiles_m + 7) / 8 * 8 * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, BK, WPS>), dim3(blocks), dim3(256), 0, 0, A, B, C, M, N, K, tiles_m, tiles_n);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE, BK, WPS>), dim3(blocks), dim3(256), 0, 0, A, B, C, M, N, K, tiles_m, tiles_n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 100759, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 1024;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    const double gf = 2.0 * M * N * K / 1e9;
#define RUN(MODE, BK, WPS) { float ms = run<MODE, BK, WPS>(A, B, C, M, N, K); printf("mode %d BK %d wps %d : %.3f ms  %.1f TFLOP/s\n", MODE, BK, WPS, ms, gf / ms); }
    RUN(0, 32, 2) RUN(1, 32, 2) RUN(2, 32, 2) RUN(3, 32, 2) RUN(4, 32, 2)
    RUN(0, 16, 3) RUN(1, 16, 3) RUN(2, 16, 3) RUN(3, 16, 3) RUN(4, 16, 3)
    RUN(0, 16, 2) RUN(4, 16, 2)
    return 0;
}
