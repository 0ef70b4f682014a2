// Which physical CUs does a CU-masked stream use on this part?  Launches a marker kernel on streams created with
// hipExtStreamCreateWithCUMask under several mask patterns and prints, per pattern, the distinct (XCC, SE, CU) triples that
// ran workgroups and the time of a fixed amount of ALU work (so that "half the CUs" shows up as ~2x the time).
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe.bin && ./tools/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void marker(uint32_t* out, int iters)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 0.5f);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 0xf) | (a == 123.f ? 16u : 0u); }
}

int main()
{
    const int NB = 8192;
    uint32_t* d; CK(hipMalloc(&d, NB * 8));
    std::vector<uint32_t> h(NB * 2);
    struct Pat { const char* name; std::vector<uint32_t> m; };
    std::vector<Pat> pats;
    pats.push_back({"none (plain stream)", {}});
    pats.push_back({"all 256 bits", std::vector<uint32_t>(8, 0xffffffffu)});
    pats.push_back({"bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}});
    pats.push_back({"bits 0..191", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0}});
    pats.push_back({"even bits", std::vector<uint32_t>(8, 0x55555555u)});
    pats.push_back({"low 16 of every 32", std::vector<uint32_t>(8, 0x0000ffffu)});
    pats.push_back({"low 24 of every 32", std::vector<uint32_t>(8, 0x00ffffffu)});
    for (auto& p : pats) {
        hipStream_t st;
        if (p.m.empty()) CK(hipStreamCreate(&st));
        else {
            hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)p.m.size(), p.m.data());
            if (e != hipSuccess) { printf("%-22s: create failed: %s\n", p.name, hipGetErrorString(e)); continue; }
        }
        CK(hipMemsetAsync(d, 0xff, NB * 8, st));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(marker, dim3(NB), dim3(256), 0, st, d, 20000);     // warm-up
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL(marker, dim3(NB), dim3(256), 0, st, d, 20000);
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost));
        std::map<int, std::set<uint32_t>> per_xcc;
        for (int i = 0; i < NB; ++i) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
            per_xcc[(int)xcc].insert((hw >> 8) & 0xff);
        }
        int tot = 0;
        printf("%-22s: %7.3f ms; CUs per XCC:", p.name, ms);
        for (auto& kv : per_xcc) { printf(" x%d=%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf("  total %d\n", tot);
        CK(hipStreamDestroy(st));
    }
    return 0;
}
