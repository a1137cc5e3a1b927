// Microbenchmark (GPU box), round 5: what can two waves that share a SIMD issue together on gfx950?
//
// The fused rollout at 131 072 envs puts exactly two waves on every SIMD (tools/wg_census.py), yet the pair retires one VALU
// instruction per ~3.4 cycles where four co-resident waves of the same kernel reach 2.05 and two waves of a pure v_fma_f32
// stream interleave perfectly (tools/micro/two_chains.hip). So the question is by instruction class: wave A runs class X,
// wave B class Y on the same SIMD — which pairs overlap, which serialise?
//
// Geometry: 512-thread workgroups, one per CU (grid = 256): eight waves, two per SIMD; the wave's HW_ID is recorded so the
// host pairs waves by the SIMD they actually ran on. Each wave runs `iters` x 256 instructions of its class (inline asm,
// eight rotating registers: dependency distance 8; a loop of its own per class) between s_memtime / s_memrealtime stamps.
// Modes: pair (X on the first wave of a SIMD, Y on the second), solo (second wave exits at once), quad (1024-thread
// workgroups: four waves per SIMD, all class X).
//   hipcc --offload-arch=gfx950 -O3 valu_pairs.hip -o /tmp/valu_pairs && /tmp/valu_pairs
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

enum Class { FMA = 0, FMAC, MUL, ADD, MOV, XOR, LSHR, ADDU, MULLO, CVT, CNDMASK, CMP, MED3, FMA_SGPR, FMA_LIT, XOR_SDWA, FMA_2SGPR, MUL_SGPR, FMA_DEP, CND_SMASK, CMP_E64, MAX3, MULHI, MUL24, CVT_I, RCP, SQRT, BFE, LSHLADD, FMA_S0, FMA_S2, FMA_NEG, FMAC_S0, MUL64_S0, FMA_INL, FMA_NEGS0, NCLASS };
static const char* kNames[NCLASS] = {"fma", "fmac_e32", "mul", "add", "mov", "xor", "lshr", "add_u32", "mul_lo_u32", "cvt_f32_u32", "cndmask", "cmp_lt",
                                     "med3", "fma_sgpr", "mul_literal", "xor_sdwa", "fma_same_sgpr_x2", "mul_sgpr_e32", "fma_dependent", "cndmask_e64_smask", "cmp_e64_sdst", "max3", "mul_hi_u32", "mul_u32_u24", "cvt_f32_i32", "rcp", "sqrt", "bfe_u32", "lshl_add_u32", "fma_sgpr_src0", "fma_sgpr_src2", "fma_neg_vgpr", "fmac_e32_sgpr_src0", "mul_e64_sgpr_src0", "fma_inline_2.0", "fma_neg_sgpr_src0"};

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R128(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

// one class = one asm statement of 128 instructions over r[0..7] with constants c1, c2 (VGPRs) and s1 (SGPR)
#define BODY(TEXT)                                                                                                                   \
    for (int i = 0; i < iters; i++)                                                                                                  \
    asm volatile(TEXT TEXT : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c1), "v"(c2), "s"(s1) : "vcc", "s20", "s21", "s22", "s23")

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_FMAC(k) "v_fmac_f32_e32 %" #k ", %8, %9\n"
#define I_MUL(k) "v_mul_f32_e32 %" #k ", %8, %" #k "\n"
#define I_ADD(k) "v_add_f32_e32 %" #k ", %8, %" #k "\n"
#define I_MOV(k) "v_mov_b32_e32 %" #k ", %8\n"
#define I_XOR(k) "v_xor_b32_e32 %" #k ", %8, %" #k "\n"
#define I_LSHR(k) "v_lshrrev_b32_e32 %" #k ", 1, %" #k "\n"
#define I_ADDU(k) "v_add_u32_e32 %" #k ", %8, %" #k "\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_CVT(k) "v_cvt_f32_u32_e32 %" #k ", %" #k "\n"
#define I_CND(k) "v_cndmask_b32_e32 %" #k ", %" #k ", %8, vcc\n"
#define I_CMP(k) "v_cmp_lt_f32_e32 vcc, %" #k ", %8\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_FMAS(k) "v_fma_f32 %" #k ", %" #k ", %10, %9\n"
#define I_FMAL(k) "v_mul_f32_e32 %" #k ", 0x3f800001, %" #k "\n"
#define I_XSDWA(k) "v_xor_b32_sdwa %" #k ", %8, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define I_FMA2S(k) "v_fma_f32 %" #k ", %" #k ", %10, %10\n"
#define I_MULS(k) "v_mul_f32_e32 %" #k ", %10, %" #k "\n"
#define I_FMAD(k) "v_fma_f32 %0, %0, %8, %9\n"  // every instruction waits for the one before it
#define I_CNDS(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n"
#define I_CMP64(k) "v_cmp_lt_f32_e64 s[22:23], %" #k ", %8\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MULHI(k) "v_mul_hi_u32 %" #k ", %" #k ", %8\n"
#define I_MUL24(k) "v_mul_u32_u24_e32 %" #k ", %8, %" #k "\n"
#define I_CVTI(k) "v_cvt_f32_i32_e32 %" #k ", %" #k "\n"
#define I_RCP(k) "v_rcp_f32_e32 %" #k ", %" #k "\n"
#define I_SQRT(k) "v_sqrt_f32_e32 %" #k ", %" #k "\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 8\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8\n"
#define I_FMAS0(k) "v_fma_f32 %" #k ", %10, %" #k ", %9\n"
#define I_FMAS2(k) "v_fma_f32 %" #k ", %" #k ", %8, %10\n"
#define I_FMANEG(k) "v_fma_f32 %" #k ", -%" #k ", %8, %9\n"
#define I_FMACS0(k) "v_fmac_f32_e32 %" #k ", %10, %8\n"
#define I_MUL64S0(k) "v_mul_f32_e64 %" #k ", %10, %" #k "\n"
#define I_FMAINL(k) "v_fma_f32 %" #k ", %" #k ", 2.0, %9\n"
#define I_FMANEGS0(k) "v_fma_f32 %" #k ", -%10, %" #k ", %9\n"

__global__ __launch_bounds__(1024) void k(int clsA, int clsB, int iters, unsigned long long* stamps, float* sink, float fc1, float fc2) {
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    // waves w and w + 4 (and + 8, + 12) of a workgroup share a SIMD: the first one of each SIMD runs class A, the others class B
    // (the class is wave-uniform: say so, or every case below becomes an exec-masked region behind a tree of vector compares)
    const int cls = __builtin_amdgcn_readfirstlane(wave < 4 ? clsA : clsB);
    float r0 = 1.0f + 1e-3f * threadIdx.x, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
    const float c1 = fc1 + 1e-9f * threadIdx.x, c2 = fc2;
    float s1 = __builtin_amdgcn_readfirstlane(fc1);
    unsigned long long* row = stamps + (size_t)(blockIdx.x * waves + wave) * 8;
    if (cls < 0) {
        if ((threadIdx.x & 63) == 0) row[4] = ~0ull;
        return;
    }
    asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555" ::: "s20", "s21");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    // one loop per class (a taken branch restarts the wave's instruction fetch: the first version of this file jumped through a
    // tree of them every 128 instructions and measured 7 cycles per instruction for a lone wave); 256 instructions per iteration
    switch (cls) {
        case FMA: BODY(R128(I_FMA)); break;
        case FMAC: BODY(R128(I_FMAC)); break;
        case MUL: BODY(R128(I_MUL)); break;
        case ADD: BODY(R128(I_ADD)); break;
        case MOV: BODY(R128(I_MOV)); break;
        case XOR: BODY(R128(I_XOR)); break;
        case LSHR: BODY(R128(I_LSHR)); break;
        case ADDU: BODY(R128(I_ADDU)); break;
        case MULLO: BODY(R128(I_MULLO)); break;
        case CVT: BODY(R128(I_CVT)); break;
        case CNDMASK: BODY(R128(I_CND)); break;
        case CMP: BODY(R128(I_CMP)); break;
        case MED3: BODY(R128(I_MED3)); break;
        case FMA_SGPR: BODY(R128(I_FMAS)); break;
        case FMA_LIT: BODY(R128(I_FMAL)); break;
        case XOR_SDWA: BODY(R128(I_XSDWA)); break;
        case FMA_2SGPR: BODY(R128(I_FMA2S)); break;
        case MUL_SGPR: BODY(R128(I_MULS)); break;
        case FMA_DEP: BODY(R128(I_FMAD)); break;
        case CND_SMASK: BODY(R128(I_CNDS)); break;
        case CMP_E64: BODY(R128(I_CMP64)); break;
        case MAX3: BODY(R128(I_MAX3)); break;
        case MULHI: BODY(R128(I_MULHI)); break;
        case MUL24: BODY(R128(I_MUL24)); break;
        case CVT_I: BODY(R128(I_CVTI)); break;
        case RCP: BODY(R128(I_RCP)); break;
        case SQRT: BODY(R128(I_SQRT)); break;
        case BFE: BODY(R128(I_BFE)); break;
        case LSHLADD: BODY(R128(I_LSHLADD)); break;
        case FMA_S0: BODY(R128(I_FMAS0)); break;
        case FMA_S2: BODY(R128(I_FMAS2)); break;
        case FMA_NEG: BODY(R128(I_FMANEG)); break;
        case FMAC_S0: BODY(R128(I_FMACS0)); break;
        case MUL64_S0: BODY(R128(I_MUL64S0)); break;
        case FMA_INL: BODY(R128(I_FMAINL)); break;
        case FMA_NEGS0: BODY(R128(I_FMANEGS0)); break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if ((threadIdx.x & 63) == 0) {
        row[0] = t0; row[1] = t1; row[2] = q0; row[3] = q1;
        row[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        row[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
        row[6] = (unsigned long long)cls;
    }
}

struct Result {
    double cycA, cycB, ghz, wall_us, frac_paired;
};

static double median(std::vector<double>& v) {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static Result run(int clsA, int clsB, int threads, int iters, unsigned long long* d_st, float* d_sink, std::vector<unsigned long long>& h) {
    const int grid = 256, waves = threads / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) k<<<grid, threads>>>(clsA, clsB, iters, d_st, d_sink, 1.0000001f, 1e-7f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<grid, threads>>>(clsA, clsB, iters, d_st, d_sink, 1.0000001f, 1e-7f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * 8 * grid * waves, hipMemcpyDeviceToHost);
    std::vector<double> a, b, clk;
    std::map<unsigned long long, int> per_simd;
    for (int w = 0; w < grid * waves; w++) {
        const unsigned long long* r = &h[(size_t)w * 8];
        if (r[4] == ~0ull) continue;
        const double cyc = double(r[1] - r[0]) / (double(iters) * 256.0);
        (int(r[6]) == clsA && (w % waves) < 4 ? a : b).push_back(cyc);
        clk.push_back(double(r[1] - r[0]) / double(r[3] - r[2]) * 0.1);
        per_simd[((r[5] & 0xF) << 32) | (r[4] & 0xFF30)] += 1;  // xcc | se sh cu simd
    }
    int paired = 0;
    for (auto& kv : per_simd) paired += kv.second == (clsB < 0 ? 1 : waves / 4) ? 1 : 0;
    Result res{median(a), median(b), median(clk), ms * 1e3, per_simd.empty() ? 0.0 : double(paired) / per_simd.size()};
    hipEventDestroy(e0); hipEventDestroy(e1);
    return res;
}

int main() {
    const int iters = 200;
    unsigned long long* d_st;
    float* d_sink;
    hipMalloc(&d_st, sizeof(unsigned long long) * 8 * 256 * 16);
    hipMalloc(&d_sink, sizeof(float) * 256 * 1024);
    std::vector<unsigned long long> h((size_t)8 * 256 * 16);
    // warm the clock up
    for (int i = 0; i < 50; i++) run(FMA, FMA, 512, iters, d_st, d_sink, h);
    printf("# cycles per instruction PER WAVE (s_memtime; median over waves); 2 waves per SIMD ideal = 4.0 each (2.0 per SIMD)\n");
    printf("# solo: one wave per SIMD\n");
    for (int c = 0; c < NCLASS; c++) {
        Result r = run(c, -1, 512, iters, d_st, d_sink, h);
        printf("solo %-18s %.2f cyc/inst  clock %.2f GHz  simds as expected %.2f\n", kNames[c], r.cycA, r.ghz, r.frac_paired);
    }
    printf("# same class on both waves of a SIMD, and four waves per SIMD\n");
    for (int c = 0; c < NCLASS; c++) {
        Result r2 = run(c, c, 512, iters, d_st, d_sink, h);
        Result r4 = run(c, c, 1024, iters, d_st, d_sink, h);
        printf("same %-18s 2 waves: first %.2f second %.2f -> %.2f per SIMD (clock %.2f, placed %.2f) | 4 waves: first %.2f others %.2f -> kernel %.1f us = %.2f cyc/inst/SIMD\n",
               kNames[c], r2.cycA, r2.cycB, 1.0 / (1.0 / r2.cycA + 1.0 / r2.cycB), r2.ghz, r2.frac_paired, r4.cycA, r4.cycB, r4.wall_us,
               r4.wall_us * 1e-6 * r4.ghz * 1e9 / (4.0 * iters * 256.0));
    }
    printf("# pairs: row = class of the first (older) wave of the SIMD, column = class of the second; entry = first/second cycles per instruction\n");
    const int sub[] = {FMA, FMAC, MUL, MOV, XOR, ADDU, MULLO, CVT, CND_SMASK, CMP_E64, MED3, FMA_SGPR, XOR_SDWA, FMA_DEP};
    const int ns = sizeof(sub) / sizeof(sub[0]);
    printf("%-14s", "");
    for (int j = 0; j < ns; j++) printf(" %-11.11s", kNames[sub[j]]);
    printf("\n");
    for (int i = 0; i < ns; i++) {
        printf("%-14.14s", kNames[sub[i]]);
        for (int j = 0; j < ns; j++) {
            Result r = run(sub[i], sub[j], 512, iters, d_st, d_sink, h);
            printf(" %4.1f/%-5.1f ", r.cycA, r.cycB);
        }
        printf("\n");
    }
    return 0;
}
