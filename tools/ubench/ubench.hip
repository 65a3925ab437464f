// ubench.hip -- micro-benchmarks behind the walker's design (tools only, not part of the library):
//   valu   : issue rate of v_fma_f32 / v_fma_mix_f32 / v_max3_f32 / SDWA cvt, 5 waves per SIMD
//   gather : divergent 16-byte loads per clock per CU from tables of several sizes, dependent chains
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float *out, uint32_t iters, float a, float b, uint32_t w)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    uint32_t h = w + threadIdx.x;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) {
                asm volatile("v_fma_mix_f32 %0, %10, %8, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %10, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %2, %10, %8, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %10, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %4, %10, %8, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %10, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %6, %10, %8, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %10, %8, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(h));
            } else if (KIND == 2) {
                asm volatile("v_max3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                             "v_max3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else {
                asm volatile("v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_u32_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_u32_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_u32_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_u32_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(h), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// Dependent gathers: every lane follows its own chain through a table of `n` 32-byte records; LOADS = 16-byte loads per step
// (1: the first half of the record, 2: both halves), the next index comes out of the loaded words.
// lanes of every group of GROUP consecutive lanes follow the same chain (same addresses in the same instruction)
template <int GROUP>
__global__ __launch_bounds__(256) void k_gather_group(const uint4 *__restrict__ table, uint32_t mask, uint32_t steps, uint32_t *out)
{
    uint32_t idx = ((blockIdx.x * 256u + threadIdx.x) / GROUP) * 2654435761u;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        const uint32_t i = idx & mask;
        const uint4 a = table[(size_t)i * 2];
        acc += a.y;
        idx = idx * 1664525u + a.x + 1013904223u;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

template <int LOADS, bool LDS_TOP>
__global__ __launch_bounds__(256) void k_gather(const uint4 *__restrict__ table, uint32_t mask, uint32_t steps, uint32_t ldsMask, uint32_t *out)
{
    __shared__ uint4 top[LDS_TOP ? 2048 : 1];
    if (LDS_TOP) {
        for (uint32_t i = threadIdx.x; i < 2048; i += 256) top[i] = table[i];
        __syncthreads();
    }
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        const uint32_t i = idx & mask;
        uint4 a, b = make_uint4(0, 0, 0, 0);
        if (LDS_TOP && (idx & ldsMask) == 0) {   // some of the steps are served from LDS
            a = top[(i >> 1) & 2047u];
        } else {
            a = table[(size_t)i * 2];
            if (LOADS == 2) b = table[(size_t)i * 2 + 1];
        }
        acc += a.y ^ b.z;
        idx = idx * 1664525u + a.x + b.w + 1013904223u;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

// One wave alone on the chip: shader clocks per step of a dependent 16-byte gather with `lanes` active lanes, each
// following its own chain through the table (how long a step of a lone walker wave takes by the lanes that still walk).
__global__ __launch_bounds__(64) void k_lone_wave(const uint4 *__restrict__ table, uint32_t mask, uint32_t steps, uint32_t lanes, uint32_t seed, unsigned long long *clocks, uint32_t *out)
{
    uint32_t idx = (threadIdx.x + 1u + seed) * 2654435761u;     // another chain every launch: nothing of it is in a cache yet
    uint32_t acc = 0;
    const unsigned long long t0 = clock64();
    if (threadIdx.x < lanes) {
        for (uint32_t s = 0; s < steps; ++s) {
            const uint4 a = table[(size_t)(idx & mask) * 2];
            acc += a.y;
            idx = idx * 1664525u + a.x + 1013904223u;
        }
    }
    const unsigned long long t1 = clock64();
    out[threadIdx.x] = acc + idx;
    if (threadIdx.x == 0) *clocks = t1 - t0;
}

static double run(void (*launch)(void *), void *arg)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(arg);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(arg);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'l') {          // ubench lone: a lone wave's step time by active lanes and table size
        unsigned long long *clk;
        uint32_t *o;
        CHECK(hipMalloc(&clk, 8));
        CHECK(hipMalloc(&o, 64 * 4));
        const size_t lsizes[] = {256u << 10, 2u << 20, 16u << 20, 64u << 20, 512u << 20};
        for (size_t bytes : lsizes) {
            const uint32_t n = (uint32_t)(bytes / 32);
            std::vector<uint32_t> h((size_t)n * 8);
            uint32_t s = 12345;
            for (auto &x : h) { s = s * 1103515245u + 12345u; x = s; }
            uint4 *table;
            CHECK(hipMalloc(&table, bytes));
            CHECK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
            printf("lone wave, table %7zu KB: clocks per step with 1 / 2 / 4 / 8 / 16 / 32 / 64 lanes:", bytes >> 10);
            for (uint32_t lanes = 1; lanes <= 64; lanes *= 2) {
                unsigned long long c = 0;
                static uint32_t seed = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    seed += 977u;
                    hipLaunchKernelGGL(k_lone_wave, dim3(1), dim3(64), 0, 0, table, n - 1, 512u, lanes, seed, clk, o);
                    CHECK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
                }
                printf(" %llu", c / 512ull);
            }
            printf("\n");
            CHECK(hipFree(table));
        }
        return 0;
    }
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const uint32_t blocks = cus * 5;          // 5 blocks of 4 waves per CU = 5 waves per SIMD
    float *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const uint32_t iters = 4096;
    const char *names[4] = {"v_fma_f32", "v_fma_mix_f32", "v_max3/min3_f32", "v_cvt_f32_u32_sdwa"};
    for (int kind = 0; kind < 4; ++kind) {
        struct A { float *out; uint32_t blocks, iters; int kind; } a = {out, blocks, iters, kind};
        const double ms = run([](void *p) {
            A *a = (A *)p;
            if (a->kind == 0) hipLaunchKernelGGL(k_valu<0>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0001f, 0.5f, 0x3C003C00u);
            if (a->kind == 1) hipLaunchKernelGGL(k_valu<1>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0001f, 0.5f, 0x3C003C00u);
            if (a->kind == 2) hipLaunchKernelGGL(k_valu<2>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0001f, 0.5f, 0x3C003C00u);
            if (a->kind == 3) hipLaunchKernelGGL(k_valu<3>, dim3(a->blocks), dim3(256), 0, 0, a->out, a->iters, 1.0001f, 0.5f, 0x3C003C00u);
        }, &a);
        const double insts = (double)blocks * 4 * iters * 64;       // wave-instructions
        printf("valu %-22s %8.3f ms  %.1f G wave-inst/s  (%.2f clocks per inst per SIMD at 2.4 GHz)\n", names[kind], ms, insts / ms / 1e6,
               ms * 1e-3 * 2.4e9 * cus * 4 / insts);
    }
    // gathers
    uint32_t *gout;
    CHECK(hipMalloc(&gout, (size_t)blocks * 256 * 4));
    const size_t sizes[] = {16u << 10, 1u << 20, 16u << 20, 64u << 20, 512u << 20};
    for (size_t bytes : sizes) {
        const uint32_t n = (uint32_t)(bytes / 32);
        std::vector<uint32_t> h((size_t)n * 8);
        uint32_t s = 12345;
        for (auto &x : h) { s = s * 1103515245u + 12345u; x = s; }
        uint4 *table;
        CHECK(hipMalloc(&table, bytes));
        CHECK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
        for (int variant = 0; variant < 4; ++variant) {
            struct G { const uint4 *t; uint32_t mask, steps, blocks; uint32_t *out; int variant; } g = {table, n - 1, 2048, blocks, gout, variant};
            const double ms = run([](void *p) {
                G *g = (G *)p;
                if (g->variant == 0) hipLaunchKernelGGL((k_gather<1, false>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, 0u, g->out);
                if (g->variant == 1) hipLaunchKernelGGL((k_gather<2, false>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, 0u, g->out);
                if (g->variant == 2) hipLaunchKernelGGL((k_gather<2, true>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, 1u, g->out);
                if (g->variant == 3) hipLaunchKernelGGL((k_gather<2, true>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, 0xFFFFFFFFu, g->out);
            }, &g);
            const double laneSteps = (double)blocks * 256 * 2048;
            const char *vn[4] = {"1 x 16 B", "2 x 16 B (one record)", "2 x 16 B, half the steps LDS", "all steps from global, LDS code present"};
            printf("gather table %6zu KB  %-40s %8.3f ms  %.2f lane-steps per clock per CU\n", bytes >> 10, vn[variant], ms,
                   laneSteps / (ms * 1e-3 * 2.4e9 * cus));
        }
        for (int grp = 0; grp < 4; ++grp) {
            struct G { const uint4 *t; uint32_t mask, steps, blocks; uint32_t *out; int grp; } g = {table, n - 1, 2048, blocks, gout, grp};
            const double ms = run([](void *p) {
                G *g = (G *)p;
                if (g->grp == 0) hipLaunchKernelGGL((k_gather_group<2>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, g->out);
                if (g->grp == 1) hipLaunchKernelGGL((k_gather_group<4>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, g->out);
                if (g->grp == 2) hipLaunchKernelGGL((k_gather_group<16>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, g->out);
                if (g->grp == 3) hipLaunchKernelGGL((k_gather_group<64>), dim3(g->blocks), dim3(256), 0, 0, g->t, g->mask, g->steps, g->out);
            }, &g);
            const int gs[4] = {2, 4, 16, 64};
            printf("gather table %6zu KB  1 x 16 B, groups of %2d lanes share the address %8.3f ms  %.2f lane-steps per clock per CU\n", bytes >> 10, gs[grp], ms,
                   (double)blocks * 256 * 2048 / (ms * 1e-3 * 2.4e9 * cus));
        }
        CHECK(hipFree(table));
    }
    return 0;
}
