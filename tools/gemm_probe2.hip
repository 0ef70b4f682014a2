// Tile-shape probe for the fp32 MFMA NT GEMM (not part of the library): 128x128 (4 waves x 64x64) vs 256x128 (4 waves x 128x64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int BM, int WPS>      // BM = 128 or 256; BN = 128; 4 waves as 2 (rows) x 2 (cols); wave tile (BM/2) x 64
__global__ __launch_bounds__(256, WPS)
void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int tiles_m, int tiles_n)
{
    constexpr int BN = 128, BK = 16, LDT = BK + 4, KQ = BK / 4, RPP = 256 / KQ, NPA = BM / RPP, NPB = BN / RPP, NMI = BM / 64;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];
    float* As = smem; float* Bs = smem + 2 * BM * LDT;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int tn = slot % tiles_n, tm = (slot / tiles_n) * 8 + xcd;
    if (tm >= tiles_m) return;
    const int t = threadIdx.x, lr = t / KQ, kq = (t % KQ) * 4;
    const int row_base = tm * BM, col_base = tn * BN;
    const float* arow[NPA]; const float* brow[NPB];
#pragma unroll
    for (int p = 0; p < NPA; ++p) arow[p] = A + (size_t)min(row_base + lr + RPP * p, M - 1) * K;
#pragma unroll
    for (int p = 0; p < NPB; ++p) brow[p] = B + (size_t)min(col_base + lr + RPP * p, N - 1) * K;
    const int wave = t >> 6, lane = t & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[NMI][2];
    for (int a = 0; a < NMI; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 ra4[NPA], rb4[NPB];
    auto g_load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) ra4[p] = ldg4(arow[p] + k0 + kq);
#pragma unroll
        for (int p = 0; p < NPB; ++p) rb4[p] = ldg4(brow[p] + k0 + kq);
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) stg4(As + buf * BM * LDT + (lr + RPP * p) * LDT + kq, ra4[p]);
#pragma unroll
        for (int p = 0; p < NPB; ++p) stg4(Bs + buf * BN * LDT + (lr + RPP * p) * LDT + kq, rb4[p]);
    };
    const int nk = K / BK;
    g_load(0); s_store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) g_load((kt + 1) * BK);
        const float* ab = As + cur * BM * LDT + (wm * (BM / 2) + l31) * LDT + 4 * h;
        const float* bb = Bs + cur * BN * LDT + (wn * 64 + l31) * LDT + 4 * h;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            float4 a4[NMI], b4[2];
#pragma unroll
            for (int i = 0; i < NMI; ++i) a4[i] = ldg4(ab + i * 32 * LDT + kg * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) b4[j] = ldg4(bb + j * 32 * LDT + kg * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < NMI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float av = q == 0 ? a4[i].x : q == 1 ? a4[i].y : q == 2 ? a4[i].z : a4[i].w;
                        const float bv = q == 0 ? b4[j].x : q == 1 ? b4[j].y : q == 2 ? b4[j].z : b4[j].w;
                        acc[i][j] = mfma32(av, bv, acc[i][j]);
                    }
        }
        if (kt + 1 < nk) s_store(cur ^ 1);
        __syncthreads();
    }
    for (int mi = 0; mi < NMI; ++mi) for (int ni = 0; ni < 2; ++ni) for (int r = 0; r < 16; ++r) {
        const int row = row_base + wm * (BM / 2) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = col_base + wn * 64 + ni * 32 + l31;
        if (row < M && col < N) C[(size_t)row * N + col] = acc[mi][ni][r];
    }
}

template <int BM, int WPS>
float run(const float* A, const float* B, float* C, int M, int N, int K)
{
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + 127) / 128;
    const int blocks = (tiles_m + 7) / 8 * 8 * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<BM, WPS>), dim3(blocks), dim3(256), 0, 0, A, B, C, M, N, K, tiles_m, tiles_n);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<BM, WPS>), dim3(blocks), dim3(256), 0, 0, A, B, C, M, N, K, tiles_m, tiles_n);
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
#define RUN(BM, WPS) { float ms = run<BM, WPS>(A, B, C, M, N, K); printf("tile %dx128 wps %d : %.3f ms  %.1f TFLOP/s\n", BM, WPS, ms, gf / ms); }
    RUN(128, 3) RUN(128, 2) RUN(256, 2) RUN(256, 1)
    return 0;
}
