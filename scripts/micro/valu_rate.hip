// valu_rate.hip -- what does ONE wave64 VALU instruction cost a SIMD of gfx950, per opcode?
//
// k_scan_region is bound by VALU issue (profiles/r06_experiments.md): its time follows its VALU instruction count at 4 cycles per
// wave-instruction and SIMD -- SQ_ACTIVE_INST_VALU (quad-cycles) equals SQ_INSTS_VALU, and dummy instructions added to the walk cost
// their full issue time.  MI355X_MICROARCH.md prices a wave64 v_fma_f32 at 2 cycles; this program measures the opcodes the scan is
// made of (and the candidates for cheaper address arithmetic): N independent chains per lane, 16 / 8 / 4 waves per CU, cycles per
// wave-instruction per SIMD from the slope of the wall time between two iteration counts.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/valu_rate.hip -o scripts/micro/valu_rate && scripts/micro/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kChains = 8;   // independent dependency chains per lane
constexpr int kUnroll = 8;   // instructions per chain and loop trip

#define OP_LIST(X)                                                                                          \
    X(0, "v_add_u32", "v_add_u32_e32 %0, %1, %0")                                                           \
    X(1, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2")                                                           \
    X(2, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1")                                                  \
    X(3, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2")                                                         \
    X(4, "v_add_u32_sdwa sext word", "v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
    X(5, "v_cndmask_b32 (vcc)", "v_cndmask_b32_e32 %0, %0, %1, vcc")                                        \
    X(6, "v_cndmask_b32 (sgpr pair)", "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")                             \
    X(7, "v_cmp_gt_u16 -> sgpr pair", "v_cmp_gt_u16_e64 s[10:11], %0, %1")                                  \
    X(8, "v_cmp_le_u32 -> vcc", "v_cmp_le_u32_e32 vcc, %0, %1")                                             \
    X(9, "v_mul_i32_i24", "v_mul_i32_i24_e32 %0, %1, %0")                                                   \
    X(10, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %2")                                                  \
    X(11, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                        \
    X(12, "v_mul_i32_i24_sdwa sext byte", "v_mul_i32_i24_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(13, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2")                                                        \
    X(14, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %2")                                                    \
    X(15, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, 8, %0")                                                 \
    X(16, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1")                                                        \
    X(17, "v_dot2c_i32_i16", "v_dot2c_i32_i16_e32 %0, %1, %2")                                              \
    X(18, "v_dot2_i32_i16 (vop3p)", "v_dot2_i32_i16 %0, %1, %2, %0")                                        \
    X(19, "v_cvt_f32_u32", "v_cvt_f32_u32_e32 %0, %0")                                                      \
    X(20, "v_mul_f32", "v_mul_f32_e32 %0, %1, %0")                                                          \
    X(21, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2")                                                          \
    X(22, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %1")                                                    \
    X(23, "v_lshrrev_b64", "v_lshrrev_b64 %0, 1, %0")                                                       \
    X(24, "v_bfe_i32", "v_bfe_i32 %0, %0, 8, 8")                                                            \
    X(25, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2")                                                    \
    X(26, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0")                                            \
    X(27, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")           \
    X(28, "v_add_f32 dpp wave_shr:1", "v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")     \
    X(29, "v_readlane_b32 (to sgpr)", "v_readlane_b32 s12, %0, 3")                                          \
    X(30, "v_cndmask_b32_e64 (vcc)", "v_cndmask_b32_e64 %0, %0, %1, vcc")                                   \
    X(31, "v_cmp_le_u32 vcc + v_cndmask vcc (pair)", "v_cmp_le_u32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %0, %1, vcc") \
    X(32, "v_cmp -> s[10:11] + v_cndmask (pair)", "v_cmp_le_u32_e64 s[10:11], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]") \
    X(33, "v_cmp vcc + v_addc_co (pair)", "v_cmp_le_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc") \
    X(34, "v_sub_u32", "v_sub_u32_e32 %0, %1, %0")                                                          \
    X(35, "v_and_b32", "v_and_b32_e32 %0, %1, %0")                                                          \
    X(36, "v_xor_b32", "v_xor_b32_e32 %0, %1, %0")                                                          \
    X(37, "v_lshlrev_b32", "v_lshlrev_b32_e32 %0, 1, %0")                                                   \
    X(38, "v_ashrrev_i32", "v_ashrrev_i32_e32 %0, 1, %0")                                                   \
    X(39, "v_max_u32", "v_max_u32_e32 %0, %1, %0")                                                          \
    X(40, "v_add_f32", "v_add_f32_e32 %0, %1, %0")                                                          \
    X(41, "v_mov_b32", "v_mov_b32_e32 %0, %1")                                                              \
    X(42, "v_mul_u32_u24", "v_mul_u32_u24_e32 %0, %1, %0")                                                  \
    X(43, "v_add_u32 e64 (vop3)", "v_add_u32_e64 %0, %1, %0")                                               \
    X(44, "v_add_u32 sgpr operand", "v_add_u32_e32 %0, s12, %0")                                            \
    X(45, "v_add_co_u32 (writes vcc)", "v_add_co_u32_e32 %0, vcc, %1, %0")                                  \
    X(46, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2")                                                          \
    X(47, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 8")                                                 \
    X(48, "v_add_u32_sdwa sext byte", "v_add_u32_sdwa %0, sext(%1), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(49, "v_cvt_i32_f32", "v_cvt_i32_f32_e32 %0, %0")                                                      \
    X(50, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1")                                                  \
    X(51, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1")

template <int OP>
__global__ __launch_bounds__(1024) void k_rate(int iters, uint32_t *out)
{
    uint32_t x[kChains];
    unsigned long long x64[kChains];
    const uint32_t a = threadIdx.x * 2654435761u + 12345u, b = threadIdx.x ^ 0x5bd1e995u;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
        x[c] = a + c;
        x64[c] = ((unsigned long long)a << 32) | (b + c);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int c = 0; c < kChains; ++c) {
#define X(ID, NAME, ASM)                                                                                              \
    if constexpr (OP == ID) {                                                                                         \
        if constexpr (ID == 22 || ID == 23) asm volatile(ASM : "+v"(x64[c]) : "v"(x64[(c + 1) % kChains]));           \
        else asm volatile(ASM : "+v"(x[c]) : "v"(a), "v"(b) : "vcc", "s10", "s11", "s12");                            \
    }
                OP_LIST(X)
#undef X
            }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < kChains; ++c) acc += x[c] + (uint32_t)x64[c];
    if (acc == 0xdeadbeefu) out[0] = acc;
}

template <int OP>
double run(const char *name, int cus, double ghz, int threads, int iters, uint32_t *d_out, hipEvent_t e0, hipEvent_t e1)
{
    float ms[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        const int it = k == 0 ? iters : 3 * iters;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rate<OP>, dim3(cus), dim3(threads), 0, 0, it, d_out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float t = 0;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            ms[k] = rep == 0 ? t : std::min(ms[k], t);
        }
    }
    (void)name;
    const double inst_per_simd = 2.0 * iters * kUnroll * kChains * (threads / 64) / 4.0;
    return (ms[1] - ms[0]) * 1e6 * ghz / inst_per_simd;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    uint32_t *d_out = nullptr;
    CHECK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("device %s, %d CUs, nominal %.0f MHz; one workgroup per CU; %d independent chains per lane\n", prop.name, cus, prop.clockRate / 1000.0, kChains);
    printf("cycles per wave64 instruction per SIMD at      16 waves/CU   8 waves/CU   4 waves/CU (one per SIMD)\n");
#define X(ID, NAME, ASM)                                                                                              \
    printf("%-34s %12.2f %12.2f %12.2f\n", NAME, run<ID>(NAME, cus, ghz, 1024, iters, d_out, e0, e1), run<ID>(NAME, cus, ghz, 512, iters, d_out, e0, e1), \
           run<ID>(NAME, cus, ghz, 256, iters, d_out, e0, e1));                                                       \
    fflush(stdout);
    OP_LIST(X)
#undef X
    return 0;
}
